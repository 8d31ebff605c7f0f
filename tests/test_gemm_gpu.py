"""
The wide-channel GEMM / implicit-GEMM convolution kernels (csrc/gemm.hip, round 5) against float64 on the SAME bf16-rounded operands
(reference products on the device in float64): Linear forward / data gradient / weight + bias gradient, the fused MLP node (GELU in
the epilogues), the 3x3 "same" and 1x1 convolutions with their gradients, the batch-norm statistics from the convolution's drain and
the BatchNorm2d + LeakyReLU (+ residual) node against torch.nn.BatchNorm2d in float64; ragged sizes (rows, features off the 128 / 64
tile sizes), the split-K shapes of the deep UNETR++ stages, bit-identical reruns.
Tolerances: bf16 outputs <= 4e-3 (half a bf16 ulp is 2e-3 relative) of the reference's largest magnitude per element and <= 2e-3 in
the 2-norm; fp32 weight gradients <= 5e-4 in the 2-norm.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(got, ref):
    got, ref = got.detach().double(), ref.detach().double()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-30))


def close_bf16(got, ref, what):
    got, ref = got.detach().double(), ref.detach().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    worst = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    assert worst <= 6e-3 and rel(got, ref) <= 3e-3, f"{what}: max {worst:.2e}, 2-norm {rel(got, ref):.2e}"


def rnd(shape, dev, seed, scale=1.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=dev) * scale)


LINEAR_CASES = [
    # R, K, N, bias, res
    (1000, 128, 512, True, False),       # ragged rows
    (32768, 128, 512, False, False),     # UNETR++ stage 0 qkvv: 256 x 4 tiles
    (512, 1024, 4096, True, True),       # stage 3 qkvv: deep K -> split-K
    (2048, 192, 576, True, False),       # Swin stage 3 qkv (features off the tile sizes)
    (130, 24, 72, True, True),           # tiny, K = 24 (one partial k-block)
    (8192, 512, 256, True, True),        # out_proj with residual
]


@pytest.mark.parametrize("R,K,N,has_bias,has_res", LINEAR_CASES)
def test_linear_forward_and_gradients_vs_float64(gpu_device, R, K, N, has_bias, has_res):
    from py4cast_amd import ops_gemm as G

    dev = gpu_device
    x = rnd((R, K), dev, 1).bfloat16().requires_grad_()
    w = (rnd((N, K), dev, 2) / K ** 0.5).requires_grad_()
    b = rnd((N,), dev, 3).requires_grad_() if has_bias else None
    res = rnd((R, N), dev, 4).bfloat16().requires_grad_() if has_res else None
    dy = rnd((R, N), dev, 5).bfloat16()
    y = G.linear(x, w, b, res)
    y.backward(dy)
    xd, wd = x.detach().double(), w.detach().bfloat16().double()
    ref = xd @ wd.t()
    if has_bias:
        ref = ref + b.detach().double()
    if has_res:
        ref = ref + res.detach().double()
    close_bf16(y, ref, "y")
    dyd = dy.double()
    close_bf16(x.grad, dyd @ wd, "dx")
    assert rel(w.grad, dyd.t() @ xd) <= 5e-4
    if has_bias:
        assert rel(b.grad, dyd.sum(0)) <= 5e-4
    if has_res:
        assert torch.equal(res.grad, dy)
    # bit-identical rerun (fixed-order split-K and slab sums)
    x2, w2 = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
    y2 = G.linear(x2, w2, None if b is None else b.detach(), None if res is None else res.detach())
    y2.backward(dy)
    assert torch.equal(y2, y) and torch.equal(x2.grad, x.grad) and torch.equal(w2.grad, w.grad)


def test_linear_on_strided_rows_and_nd_input(gpu_device):
    """rows that are a column slice of a wider tensor (row stride > K) and a 4-D input"""
    from py4cast_amd import ops_gemm as G

    dev = gpu_device
    wide = rnd((2, 16, 24, 160), dev, 7).bfloat16()
    x = wide[..., 32:128]                       # K = 96, row stride 160, base offset 64 B
    w = rnd((136, 96), dev, 8) / 10
    y = G.linear(x, w)
    close_bf16(y, x.double() @ w.bfloat16().double().t(), "y")
    assert y.shape == (2, 16, 24, 136)


@pytest.mark.parametrize("R,K,Hd", [(3000, 96, 384), (512, 384, 1536), (16384, 48, 192)])
def test_mlp_node_vs_float64(gpu_device, R, K, Hd):
    from py4cast_amd import ops_gemm as G

    dev = gpu_device
    x = rnd((R, K), dev, 11).bfloat16().requires_grad_()
    w1 = (rnd((Hd, K), dev, 12) / K ** 0.5).requires_grad_()
    b1 = rnd((Hd,), dev, 13, 0.5).requires_grad_()
    w2 = (rnd((K, Hd), dev, 14) / Hd ** 0.5).requires_grad_()
    b2 = rnd((K,), dev, 15, 0.5).requires_grad_()
    dy = rnd((R, K), dev, 16).bfloat16()
    y = G.mlp(x, w1, b1, w2, b2, res=x)
    y.backward(dy)
    # reference in float64 with the roundings the node makes: pre-activation h and g = gelu(h) are stored as bf16
    xd = x.detach().double().requires_grad_()
    w1d, w2d = w1.detach().bfloat16().double().requires_grad_(), w2.detach().bfloat16().double().requires_grad_()
    b1d, b2d = b1.detach().double().requires_grad_(), b2.detach().double().requires_grad_()
    h = (xd @ w1d.t() + b1d)
    g = F.gelu(h.detach().bfloat16().double() + (h - h.detach()))            # value rounded, gradient through
    g = g.detach().bfloat16().double() + (g - g.detach())
    ref = g @ w2d.t() + b2d + xd
    ref.backward(dy.double())
    close_bf16(y, ref, "y")
    close_bf16(x.grad, xd.grad, "dx")
    assert rel(w1.grad, w1d.grad) <= 3e-3 and rel(w2.grad, w2d.grad) <= 1e-3      # dh is rounded to bf16 before the first layer's products
    assert rel(b1.grad, b1d.grad) <= 3e-3 and rel(b2.grad, b2d.grad) <= 5e-4


CONV_CASES = [
    # B, H, W, Ci, Co, k
    (2, 32, 32, 128, 128, 3),
    (2, 16, 16, 256, 256, 3),       # split-K
    (1, 20, 12, 64, 136, 3),        # ragged map, outputs off the tile width
    (2, 16, 16, 1024, 1024, 3),     # UNETR++ stage 3 (K = 9216: 8 splits)
    (2, 24, 40, 96, 48, 3),         # Swin decoder widths
    (2, 16, 16, 512, 512, 1),       # 1x1 (conv8)
    (3, 128, 128, 128, 128, 3),     # 3 x 128 tiles of rows, no split
]


def conv_ref(x, w, k):
    """float64 convolution of the bf16-rounded operands on the device: one matmul per tap"""
    B, H, W, Ci = x.shape
    Co = w.shape[0]
    xd, wd = x.double(), w.bfloat16().double()
    if k == 1:
        return (xd.reshape(-1, Ci) @ wd.reshape(Co, Ci).t()).view(B, H, W, Co)
    xp = F.pad(xd, (0, 0, 1, 1, 1, 1))
    y = torch.zeros(B, H, W, Co, dtype=torch.float64, device=x.device)
    for ky in range(3):
        for kx in range(3):
            y += (xp[:, ky:ky + H, kx:kx + W, :].reshape(-1, Ci) @ wd[:, :, ky, kx].t()).view(B, H, W, Co)
    return y


@pytest.mark.parametrize("B,H,W,Ci,Co,k", CONV_CASES)
def test_convolution_forward_and_gradients_vs_float64(gpu_device, B, H, W, Ci, Co, k):
    from py4cast_amd import ops_gemm as G

    dev = gpu_device
    x = rnd((B, H, W, Ci), dev, 21).bfloat16().requires_grad_()
    w = (rnd((Co, Ci, k, k), dev, 22) / (Ci * k * k) ** 0.5).requires_grad_()
    dy = rnd((B, H, W, Co), dev, 23).bfloat16()
    y, stats = G.conv2d_nhwc(x, w, want_stats=True)
    y.backward(dy)
    ref = conv_ref(x.detach(), w.detach(), k)
    close_bf16(y, ref, "y")
    # statistics of the ROUNDED output, from the drain
    yd = y.detach().double().reshape(-1, Co)
    s = stats.double().sum(0)
    assert float(((s[0] - yd.sum(0)).abs() / yd.abs().sum(0).clamp_min(1e-30)).max()) <= 1e-5
    assert rel(s[1], (yd * yd).sum(0)) <= 1e-5
    # data gradient = convolution of dy with the transposed, flipped kernel; weight gradient = sum over pixels
    dyd = dy.double()
    wt = w.detach().transpose(0, 1).flip(2, 3).contiguous() if k == 3 else w.detach().transpose(0, 1).contiguous()
    close_bf16(x.grad, conv_ref(dy, wt, k), "dx")
    xd = x.detach().double()
    if k == 3:
        xp = F.pad(xd, (0, 0, 1, 1, 1, 1))
        gw = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, device=dev)
        for ky in range(3):
            for kx in range(3):
                gw[:, :, ky, kx] = dyd.reshape(-1, Co).t() @ xp[:, ky:ky + H, kx:kx + W, :].reshape(-1, Ci)
    else:
        gw = (dyd.reshape(-1, Co).t() @ xd.reshape(-1, Ci)).view(Co, Ci, 1, 1)
    assert rel(w.grad, gw) <= 5e-4
    # reruns are bit-identical
    x2, w2 = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
    y2, stats2 = G.conv2d_nhwc(x2, w2, want_stats=True)
    y2.backward(dy)
    assert torch.equal(y2, y) and torch.equal(stats2, stats) and torch.equal(x2.grad, x.grad) and torch.equal(w2.grad, w.grad)


def test_convolution_with_bias_residual_and_wider_input_map(gpu_device):
    """1x1 convolution with bias + residual (UNETR++'s conv8), and a 3x3 one reading the first 64 channels of a 96-channel map"""
    from py4cast_amd import ops_gemm as G

    dev = gpu_device
    x = rnd((2, 16, 16, 256), dev, 31).bfloat16().requires_grad_()
    w = (rnd((256, 256, 1, 1), dev, 32) / 16).requires_grad_()
    b = rnd((256,), dev, 33).requires_grad_()
    skip = rnd((2, 16, 16, 256), dev, 34).bfloat16().requires_grad_()
    dy = rnd((2, 16, 16, 256), dev, 35).bfloat16()
    y = G.conv2d_nhwc(x, w, b, res=skip)
    y.backward(dy)
    close_bf16(y, conv_ref(x.detach(), w.detach(), 1) + b.detach().double() + skip.detach().double(), "y")
    assert rel(b.grad, dy.double().sum((0, 1, 2))) <= 5e-4 and torch.equal(skip.grad, dy)
    xw = rnd((1, 24, 24, 96), dev, 36).bfloat16().requires_grad_()
    w3 = (rnd((64, 64, 3, 3), dev, 37) / 24).requires_grad_()
    y3 = G.conv2d_nhwc(xw, w3)
    y3.backward(torch.ones_like(y3))
    close_bf16(y3, conv_ref(xw.detach()[..., :64], w3.detach(), 3), "y3")
    assert float(xw.grad[..., 64:].abs().max()) == 0.0


@pytest.mark.parametrize("C,slope,with_res", [(128, 0.01, False), (256, 0.01, True), (1024, 1.0, False)])
def test_batch_norm_node_vs_torch_float64(gpu_device, C, slope, with_res):
    from py4cast_amd import ops_gemm as G

    dev = gpu_device
    B, H, W = 2, 16, 16
    x = rnd((B, H, W, C // 2), dev, 41).bfloat16()
    w = rnd((C, C // 2, 3, 3), dev, 42) / (C * 4.5) ** 0.5
    bn = torch.nn.BatchNorm2d(C).to(dev)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.1 * rnd((C,), dev, 43))
        bn.bias.copy_(0.1 * rnd((C,), dev, 44))
    ref_bn = torch.nn.BatchNorm2d(C).to(dev).double()
    ref_bn.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    res = rnd((B, H, W, C), dev, 45).bfloat16().requires_grad_() if with_res else None
    dy = rnd((B, H, W, C), dev, 46).bfloat16()
    y, stats = G.conv2d_nhwc(x, w, want_stats=True)
    yl = y.detach().requires_grad_()
    out = G.batch_norm_act(yl, stats, bn, slope, res)
    out.backward(dy)
    yr = yl.detach().double().requires_grad_()
    rr = None if res is None else res.detach().double().requires_grad_()
    o = ref_bn(yr.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    if rr is not None:
        o = o + rr
    o = F.leaky_relu(o, slope) if slope != 1.0 else o
    o.backward(dy.double())
    close_bf16(out, o, "out")
    close_bf16(yl.grad, yr.grad, "dy")
    assert rel(bn.weight.grad, ref_bn.weight.grad) <= 5e-3 and rel(bn.bias.grad, ref_bn.bias.grad) <= 5e-3
    assert rel(bn.running_mean, ref_bn.running_mean) <= 1e-4 and rel(bn.running_var, ref_bn.running_var) <= 1e-4
    if with_res:
        close_bf16(res.grad, rr.grad, "dres")
    # eval mode: running statistics
    bn.eval(), ref_bn.eval()
    oe = G.batch_norm_act(y.detach(), None, bn, slope, None)
    re_ = ref_bn(y.detach().double().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    close_bf16(oe, F.leaky_relu(re_, slope) if slope != 1.0 else re_, "eval out")


def test_batch_norm_node_with_channel_dropout_and_handed_on_residual(gpu_device):
    """Round 6: (a) ``mul`` -- a multiplier per (sample, channel) behind the activation (the Dropout2d in front of UNETR++'s conv8) applied
    by the normalisation passes themselves, forward and backward, against float64 torch with the same mask; (b) ``res_passthrough`` /
    ``conv2d_nhwc(passthrough=True)`` -- a tensor with several consumers handed from one to the next, its gradients meeting inside the
    backward launches: same gradients as autograd's additions; (c) num_batches_tracked is advanced by the statistics kernel."""
    from py4cast_amd import ops_gemm as G

    dev = gpu_device
    B, H, W, C, slope = 2, 16, 16, 128, 0.01
    x = rnd((B, H, W, C), dev, 61).bfloat16().requires_grad_()
    w = torch.nn.Parameter(rnd((C, C, 3, 3), dev, 62) / (C * 9) ** 0.5)
    bn = torch.nn.BatchNorm2d(C).to(dev)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.1 * rnd((C,), dev, 63))
        bn.bias.copy_(0.1 * rnd((C,), dev, 64))
    ref_bn = torch.nn.BatchNorm2d(C).to(dev).double()
    ref_bn.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    mask = (torch.rand(B, C, device=dev, generator=torch.Generator(device=dev).manual_seed(65)) < 0.7).float()
    mask[0, :4] = 0.0                                            # (dropped channels for sure)
    factor = 1.0 / 0.7
    dy = rnd((B, H, W, C), dev, 66).bfloat16()
    dx2 = rnd((B, H, W, C), dev, 67).bfloat16()
    # x -> conv (hands x on) -> batch norm + residual x (hands it on) + dropout -> out;  x2 = the handed-on x, used once more
    y, stats, xp = G.conv2d_nhwc(x, w, want_stats=True, passthrough=True)
    out, x2 = G.batch_norm_act(y, stats, bn, slope, xp, res_passthrough=True, mul=mask, mul_factor=factor)
    assert int(bn.num_batches_tracked) == 1
    torch.autograd.backward([out, x2], [dy, dx2])
    # float64 reference on the same bf16 operands
    xr = x.detach().double().requires_grad_()
    wr = w.detach().double().requires_grad_()
    # (the normalisation's reference starts from the node's own bf16 y, as in the test above: a y computed apart flips the LeakyReLU branch
    #  of the elements nearest to zero, and every flip moves a gradient sum by a whole term)
    yr = y.detach().double().permute(0, 3, 1, 2).requires_grad_()
    o = F.leaky_relu(ref_bn(yr) + xr.permute(0, 3, 1, 2), slope) * (mask.double() * factor)[:, :, None, None]
    o = o.permute(0, 2, 3, 1)
    o.backward(dy.double())
    close_bf16(out, o, "out")
    assert float(out[0, :, :, :4].abs().max()) == 0.0
    assert rel(bn.weight.grad, ref_bn.weight.grad) <= 5e-3 and rel(bn.bias.grad, ref_bn.bias.grad) <= 5e-3
    # gradient of x: through the residual (float64: xr.grad so far) + through the convolution (dy_conv = yr.grad) + the third consumer
    g_conv = torch.autograd.grad(F.conv2d(xr.permute(0, 3, 1, 2), wr, padding=1), xr, yr.grad)[0]
    want = xr.grad + g_conv + dx2.double()
    close_bf16(x.grad, want, "dx (three consumers, no addition launch)")


def test_accumulating_weight_gradients_are_reduced_in_batches_at_the_end_of_the_backward(gpu_device):
    """Round 6: inside a backward pass the split-K slabs of the weight-gradient calls that ADD into .grad buffers are queued and summed 32
    calls per launch when the pass ends (csrc/gemm.hip: gemm_tn_reduce_batch_kernel, driven by ops_nodeproj.GradQueue).  Same bits as
    one reduction per call -- also for a weight used by several steps of a rollout (its additions stay in submission order) -- and
    nothing is left in the queue."""
    from py4cast_amd import _lib as L
    from py4cast_amd import ops_gemm as G
    from py4cast_amd.ops_nodeproj import GradQueue

    dev = gpu_device
    torch.manual_seed(71)
    lins = [torch.nn.Linear(128, 256).to(dev), torch.nn.Linear(256, 128).to(dev), torch.nn.Linear(128, 128, bias=False).to(dev)]
    conv = torch.nn.Conv2d(128, 128, 3, padding=1, bias=False).to(dev)
    params = [p for m in lins + [conv] for p in m.parameters()]
    x0 = rnd((2, 16, 16, 128), dev, 72).bfloat16()

    def rollout():
        x, loss = x0, 0.0
        for _ in range(3):                                   # three "AR steps" on the same weights
            h = G.linear(x, lins[0].weight, lins[0].bias)
            h = G.linear(h, lins[1].weight, lins[1].bias)
            h = G.conv2d_nhwc(h, conv.weight)
            x = G.linear(h, lins[2].weight, None, res=x)
            loss = loss + x.float().square().mean()
        return loss

    out = {}
    for mode in (False, True, True):
        for p in params:
            p.grad = torch.zeros_like(p)
        GradQueue.enabled = mode
        try:
            rollout().backward()
        finally:
            GradQueue.enabled = True
        assert L.lib().p4c_grad_reduce_pending() == 0
        out.setdefault(mode, []).append([p.grad.clone() for p in params])
    for a, b in zip(out[False][0], out[True][0]):
        assert float(a.abs().max()) > 0 and torch.equal(a, b)
    for a, b in zip(out[True][0], out[True][1]):
        assert torch.equal(a, b)


def test_all_stale_weight_images_are_prepared_by_the_first_miss_of_a_step(gpu_device):
    """Round 6: the first lookup that misses prepares the images of every weight used before whose cache entry is stale, 24 per launch
    (p4c_gemm_prep_weight_batch) -- the same bits as one preparation per weight, for plain and layer-scaled weights, and the later
    lookups of the step launch nothing."""
    from py4cast_amd import _lib as L
    from py4cast_amd import ops_gemm as G

    dev = gpu_device
    torch.manual_seed(81)
    ws = [torch.nn.Parameter(rnd(shape, dev, 82 + i)) for i, shape in enumerate([(64, 32, 3, 3), (128, 64), (96, 96, 3, 3), (40, 72), (256, 128)])]
    taps = [9, 1, 9, 1, 1]
    gamma = torch.nn.Parameter(1 + 0.1 * rnd((256 + 128,), dev, 90))
    sw = [torch.nn.Parameter(rnd((256, 64), dev, 91)), torch.nn.Parameter(rnd((128, 64), dev, 92))]
    sb = [torch.nn.Parameter(rnd((256,), dev, 93)), None]
    sg = lambda: (gamma[:256], gamma[256:])     # noqa: E731   (fresh slices every time, as the model makes them)

    def lookup_all():
        out = [G.weight_images(w, t) for w, t in zip(ws, taps)]
        out += [G.scaled_images(w, b, g) for w, b, g in zip(sw, sb, sg())]
        return out

    G._WIMG.clear()
    G._PREP_LOG.clear()
    lookup_all()                                               # first use: logged (and prepared)
    assert len(G._PREP_LOG) == len(ws) + len(sw)
    with torch.no_grad():                                      # "the optimizer moved the parameters"
        for p in ws + sw + [gamma, sb[0]]:
            p.mul_(1.01)
    calls = []
    real = L.call
    L.call = lambda name, *a, **k: (calls.append(name), real(name, *a, **k))[1]
    try:
        batched = lookup_all()
    finally:
        L.call = real
    assert [c for c in calls if "prep" in c] == ["p4c_gemm_prep_weight_batch"], calls      # ONE launch request for all seven
    G.BATCHED_PREP = False
    try:
        G._WIMG.clear()
        single = lookup_all()
    finally:
        G.BATCHED_PREP = True
    for a, b in zip(batched, single):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert (x is None and y is None) or torch.equal(x, y)
    del ws[2]                                                  # a weight that is gone leaves the log at the next miss
    with torch.no_grad():
        ws[0].mul_(1.01)
    import gc

    gc.collect()
    G.weight_images(ws[0], 9)
    assert len(G._PREP_LOG) == len(ws) + len(sw)


def test_weight_images_follow_the_parameter_version(gpu_device):
    from py4cast_amd import ops_gemm as G

    dev = gpu_device
    w = torch.nn.Parameter(rnd((64, 32, 3, 3), dev, 51))
    f1, d1 = G.weight_images(w, 9)
    assert G.weight_images(w, 9)[0] is f1
    ref = w.detach().permute(0, 2, 3, 1).reshape(64, 9 * 32).bfloat16()
    assert torch.equal(f1, ref)
    refd = w.detach().flip(2, 3).permute(1, 2, 3, 0).reshape(32, 9 * 64).bfloat16()
    assert torch.equal(d1, refd)
    with torch.no_grad():
        w.mul_(2.0)
    f2, _ = G.weight_images(w, 9)
    assert f2 is not f1 and torch.equal(f2, (ref.float() * 2).bfloat16())


def test_no_cpu_path(gpu_device):
    from py4cast_amd import _lib as L
    from py4cast_amd import ops_gemm as G

    with pytest.raises(L.P4CError):
        G.linear(torch.zeros(4, 8, dtype=torch.bfloat16), torch.zeros(8, 8))
    with pytest.raises(L.P4CError):
        G.conv2d_nhwc(torch.zeros(1, 4, 4, 8, dtype=torch.bfloat16), torch.zeros(8, 8, 3, 3))


@pytest.mark.parametrize("B,H,W,C,scale,with_skip", [(2, 16, 16, 64, 2, True), (1, 12, 20, 128, 4, True), (2, 8, 8, 256, 2, False)])
def test_bilinear_upsample_add_vs_torch_float64(gpu_device, B, H, W, C, scale, with_skip):
    """csrc/resize.hip against F.interpolate(bilinear, align_corners=False) (+ skip) in float64: forward, and the gather-form backward
    against autograd's; bit-identical reruns"""
    from py4cast_amd import ops_gemm as G

    dev = gpu_device
    x = rnd((B, H, W, C), dev, 61).bfloat16().requires_grad_()
    skip = rnd((B, H * scale, W * scale, C), dev, 62).bfloat16().requires_grad_() if with_skip else None
    dy = rnd((B, H * scale, W * scale, C), dev, 63).bfloat16()
    y = G.upsample_add(x, skip, scale)
    y.backward(dy)
    xd = x.detach().double().requires_grad_()
    ref = F.interpolate(xd.permute(0, 3, 1, 2), scale_factor=scale, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    if with_skip:
        ref = ref + skip.detach().double()
    ref.backward(dy.double())
    close_bf16(y, ref, "y")
    close_bf16(x.grad, xd.grad, "dx")
    if with_skip:
        assert torch.equal(skip.grad, dy)
    x2 = x.detach().clone().requires_grad_()
    y2 = G.upsample_add(x2, None if skip is None else skip.detach(), scale)
    y2.backward(dy)
    assert torch.equal(y2, y) and torch.equal(x2.grad, x.grad)


@pytest.mark.parametrize("R,C,N", [(512, 1024, 256), (4096, 128, 2048), (300, 520, 100)])
def test_add_layer_norm_vs_float64(gpu_device, R, C, N):
    """(x + pos) -> LayerNorm on rows up to 2 KiB (csrc/rows.hip): both outputs and every gradient against float64"""
    from py4cast_amd import ops_rows as RW

    dev = gpu_device
    B = R // N if R % N == 0 else 1
    if R % N:
        N = R
    x = rnd((B, N, C), dev, 71).bfloat16().requires_grad_()
    pos = (0.1 * rnd((1, N, C), dev, 72)).requires_grad_()
    g = (1 + 0.1 * rnd((C,), dev, 73)).requires_grad_()
    b = (0.1 * rnd((C,), dev, 74)).requires_grad_()
    dt, dln = rnd((B, N, C), dev, 75).bfloat16(), rnd((B, N, C), dev, 76).bfloat16()
    t, ln = RW.add_layer_norm(x, pos, g, b, 1e-5)
    (t.float() * dt.float()).sum().backward(retain_graph=True) if False else torch.autograd.backward([t, ln], [dt, dln])
    xd, pd = x.detach().double().requires_grad_(), pos.detach().bfloat16().double().requires_grad_()
    gd, bd = g.detach().double().requires_grad_(), b.detach().double().requires_grad_()
    tr = xd + pd
    trr = tr.detach().bfloat16().double() + (tr - tr.detach())          # the stored (rounded) sum is what is normalised
    lr = F.layer_norm(trr, (C,), gd, bd, 1e-5)
    torch.autograd.backward([tr, lr], [dt.double(), dln.double()])
    close_bf16(t, tr, "t")
    close_bf16(ln, lr, "ln")
    close_bf16(x.grad, xd.grad, "dx")
    assert rel(pos.grad, pd.grad) <= 5e-3 and rel(g.grad, gd.grad) <= 2e-3 and rel(b.grad, bd.grad) <= 2e-3


def test_weight_gradients_accumulate_into_existing_grad_buffers(gpu_device):
    """GRADS_IN_PLACE: with .grad buffers in place the reduction kernel ADDS dW / db into them (no AccumulateGrad launch); the sums
    equal the ordinary path's, for a Linear with bias, a 3x3 convolution and a parameter seen through a detached stand-in
    (trainer.RolloutParamProxies)."""
    from py4cast_amd import ops_gemm as G

    dev = gpu_device
    x = rnd((600, 64), dev, 81).bfloat16()
    dy = rnd((600, 136), dev, 82).bfloat16()
    w, b = torch.nn.Parameter(rnd((136, 64), dev, 83) / 8), torch.nn.Parameter(rnd((136,), dev, 84))
    G.linear(x, w, b).backward(dy)                       # no buffers yet: through autograd
    ref_w, ref_b = w.grad.clone(), b.grad.clone()
    w.grad.fill_(1.0), b.grad.fill_(2.0)
    G.linear(x, w, b).backward(dy)                       # buffers exist: added in place
    assert torch.allclose(w.grad, ref_w + 1.0, rtol=0, atol=1e-5 * float(ref_w.abs().max())) and torch.allclose(b.grad, ref_b + 2.0, rtol=0, atol=1e-4)
    # a stand-in of the parameter (same storage, owner attribute): the owner's buffer receives the sum
    q = w.detach().requires_grad_(True)
    q._p4c_owner = w
    w.grad.zero_()
    G.linear(x, q, None).backward(dy)
    assert q.grad is None and torch.allclose(w.grad, ref_w, rtol=0, atol=1e-5 * float(ref_w.abs().max()))
    xc = rnd((2, 16, 16, 64), dev, 85).bfloat16()
    dyc = rnd((2, 16, 16, 72), dev, 86).bfloat16()
    wc = torch.nn.Parameter(rnd((72, 64, 3, 3), dev, 87) / 24)
    G.conv2d_nhwc(xc, wc).backward(dyc)
    ref_c = wc.grad.clone()
    G.conv2d_nhwc(xc, wc).backward(dyc)
    assert torch.allclose(wc.grad, 2 * ref_c, rtol=0, atol=1e-5 * float(ref_c.abs().max()))


@pytest.mark.parametrize("R,K,Hd", [(8192, 24, 96), (16500, 48, 192)])
def test_row_gemm_mlp_node_vs_float64(gpu_device, R, K, Hd):
    """The streaming row-GEMM kernels' fused epilogues (csrc/rowgemm.hip, round 5): res + fc2(gelu(fc1(x))) as one node -- GELU / GELU'
    and the residual inside the products -- against float64 with the node's roundings (h and g stored as bf16)."""
    from py4cast_amd import ops_rows as RW

    dev = gpu_device
    x = rnd((R, K), dev, 91).bfloat16().requires_grad_()
    w1 = (rnd((Hd, K), dev, 92) / K ** 0.5).requires_grad_()
    b1 = rnd((Hd,), dev, 93, 0.5).requires_grad_()
    w2 = (rnd((K, Hd), dev, 94) / Hd ** 0.5).requires_grad_()
    b2 = rnd((K,), dev, 95, 0.5).requires_grad_()
    dy = rnd((R, K), dev, 96).bfloat16()
    assert RW.row_mlp_gelu_ok(x, w1, b1, w2, b2)
    y = RW.row_mlp_gelu(x, w1, b1, w2, b2, res=x)
    y.backward(dy)
    xd = x.detach().double().requires_grad_()
    w1d, w2d = w1.detach().bfloat16().double().requires_grad_(), w2.detach().bfloat16().double().requires_grad_()
    b1d, b2d = b1.detach().double().requires_grad_(), b2.detach().double().requires_grad_()
    h = xd @ w1d.t() + b1d
    g = F.gelu(h.detach().bfloat16().double() + (h - h.detach()))
    g = g.detach().bfloat16().double() + (g - g.detach())
    ref = g @ w2d.t() + b2d + xd
    ref.backward(dy.double())
    close_bf16(y, ref, "y")
    close_bf16(x.grad, xd.grad, "dx")
    assert rel(w1.grad, w1d.grad) <= 3e-3 and rel(w2.grad, w2d.grad) <= 1e-3 and rel(b1.grad, b1d.grad) <= 3e-3 and rel(b2.grad, b2d.grad) <= 5e-4
    # a projection with the residual in its epilogue, and accumulation into an existing tensor
    w = (rnd((K, K), dev, 97) / K ** 0.5)
    b = rnd((K,), dev, 98)
    z = RW.linear_res(x.detach(), w, b, x.detach())
    close_bf16(z, x.detach().double() @ w.bfloat16().double().t() + b.double() + x.detach().double(), "linear_res")


@pytest.mark.parametrize("R,K,O", [(1000, 128, 64), (512, 1024, 512), (4096, 256, 128)])
def test_cat_linear_with_layer_scale_vs_float64(gpu_device, R, K, O):
    """res + gamma * cat(xa Wa^T + ba, xb Wb^T + bb) (UNETR++'s `t + gamma * epa(...)`): gamma folded into the weight images by the
    preparation kernel, dW / db / dgamma from the raw gradients by p4c_gemm_scale_fold_bwd.  Against float64 on the same bf16 rows,
    through autograd (no gradient buffers) and in place (buffers exist: everything is ADDED, nothing returned)."""
    from py4cast_amd import ops_gemm as G

    dev = gpu_device
    xa, xb = rnd((R, K), dev, 91).bfloat16(), rnd((R, K), dev, 92).bfloat16()
    res, dy = rnd((R, 2 * O), dev, 93).bfloat16(), rnd((R, 2 * O), dev, 94).bfloat16()
    P = lambda t: torch.nn.Parameter(t)   # noqa: E731
    wa, wb = P(rnd((O, K), dev, 95) / K ** 0.5), P(rnd((O, K), dev, 96) / K ** 0.5)
    ba, bb, gamma = P(rnd((O,), dev, 97)), P(rnd((O,), dev, 98)), P(rnd((2 * O,), dev, 99))
    xag, xbg, resg = (t.clone().requires_grad_(True) for t in (xa, xb, res))
    y = G.cat_linear_res(xag, wa, ba, xbg, wb, bb, resg, gamma=gamma)
    y.backward(dy)
    # float64 reference with the operands the kernels see: bf16 rows, bf16(gamma * W) images, fp32 gamma * b
    d = lambda t: t.detach().double()   # noqa: E731
    xar, xbr, rr = (d(t).requires_grad_(True) for t in (xa, xb, res))
    war, wbr, bar, bbr, gr = (d(t).requires_grad_(True) for t in (wa, wb, ba, bb, gamma))
    z = torch.cat([F.linear(xar, war, bar), F.linear(xbr, wbr, bbr)], dim=-1)
    yr = rr + gr * z
    yr.backward(d(dy))
    close_bf16(y, yr, "y")
    close_bf16(xag.grad, xar.grad, "dxa")
    close_bf16(xbg.grad, xbr.grad, "dxb")
    assert torch.equal(resg.grad, dy)
    for got, ref, what in ((wa.grad, war.grad, "dWa"), (wb.grad, wbr.grad, "dWb"), (ba.grad, bar.grad, "dba"), (bb.grad, bbr.grad, "dbb"),
                           (gamma.grad, gr.grad, "dgamma")):
        # (the forward rounds gamma * W to bf16 for the images -- the gradients use the fp32 masters: 5e-4 like every weight gradient here;
        # dgamma sums products with z = x W^T of the UNROUNDED weight: same bar)
        assert rel(got, ref) <= 5e-4, (what, rel(got, ref))
    # second pass: the buffers exist -> added in place, bit-identical increments
    first = [p.grad.clone() for p in (wa, wb, ba, bb, gamma)]
    G.cat_linear_res(xa, wa, ba, xb, wb, bb, res, gamma=gamma).backward(dy)
    for p, f in zip((wa, wb, ba, bb, gamma), first):
        assert torch.allclose(p.grad, 2 * f, rtol=0, atol=2e-6 * float(f.abs().max()))
    # and the same numbers as the torch-product form it replaces (gamma * W, gamma * b as autograd nodes)
    for p in (wa, wb, ba, bb, gamma):
        p.grad = None
    h = O
    y2 = G.cat_linear_res(xa, gamma[:h].unsqueeze(1) * wa, gamma[:h] * ba, xb, gamma[h:].unsqueeze(1) * wb, gamma[h:] * bb, res)
    assert torch.equal(y2, y.detach())
    y2.backward(dy)
    for p, f, what in zip((wa, wb, ba, bb, gamma), first, ("dWa", "dWb", "dba", "dbb", "dgamma")):
        assert rel(p.grad, f) <= 1e-5, (what, rel(p.grad, f))
