"""
GPU parity of the model kernels (conv on the fp32 matrix cores, norms, pool, up-sample-and-sum, the
HalfUNet plan) against plain PyTorch fp32 references on the CPU (oracle/halfunet.py for the network).
Tolerance: 1e-4 relative (north-star bar for fp32 forward outputs); gradients 1e-3.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu


def rel_err(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


@pytest.mark.parametrize("CI,CIreal,ks,H,W", [(64, 64, 3, 16, 32), (96, 69, 3, 20, 40), (32, 10, 3, 8, 8), (64, 64, 1, 12, 36),
                                                (64, 60, 3, 4, 4), (96, 96, 1, 8, 64)])
def test_conv_fwd_matches_torch(gpu_device, CI, CIreal, ks, H, W):
    from py4cast_amd import ops_model as om

    g = torch.Generator().manual_seed(CI + ks + H)
    B, CO = 2, 64
    x = torch.randn(B, H, W, CI, generator=g)
    x[..., CIreal:] = 0
    w = torch.randn(CO, CIreal, ks, ks, generator=g) * 0.1
    scale = torch.rand(B, CI, generator=g) + 0.5
    shift = torch.randn(B, CI, generator=g) * 0.3
    for transform in (False, True):
        xin = x
        if transform:
            xin = torch.relu(x * scale[:, None, None, :] + shift[:, None, None, :])
            xin = xin.clone()
            xin[..., CIreal:] = xin[..., CIreal:]  # padded channels: weights are zero there
        ref = Fn.conv2d(xin[..., :CIreal].permute(0, 3, 1, 2), w, padding=ks // 2).permute(0, 2, 3, 1)
        wp = om.prep_weights(w.to(gpu_device), False, 64, CI)
        out, stats = om.conv_fwd(x.to(gpu_device), wp, ks, in_scale=scale.to(gpu_device) if transform else None,
                                 in_shift=shift.to(gpu_device) if transform else None, in_relu=transform, want_stats=True)
        assert rel_err(out, ref) < 1e-5
        s = stats.sum(0).cpu()
        np.testing.assert_allclose(s[0].numpy(), ref.sum((0, 1, 2)).numpy(), rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(s[1].numpy(), (ref**2).sum((0, 1, 2)).numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("CIreal,ks,H,W", [(64, 3, 16, 32), (60, 3, 12, 20), (64, 1, 8, 40)])
def test_conv_data_grad_matches_autograd(gpu_device, CIreal, ks, H, W):
    from py4cast_amd import ops_model as om

    g = torch.Generator().manual_seed(7)
    B, CO = 2, 64
    x = torch.randn(B, CIreal, H, W, generator=g, requires_grad=True)
    w = torch.randn(CO, CIreal, ks, ks, generator=g) * 0.1
    dout = torch.randn(B, H, W, CO, generator=g)
    Fn.conv2d(x, w, padding=ks // 2).backward(dout.permute(0, 3, 1, 2))
    ref = x.grad.permute(0, 2, 3, 1)
    wp = om.prep_weights(w.to(gpu_device), True, 64, 64)
    got = om.conv_fwd(dout.to(gpu_device), wp, ks)
    assert rel_err(got[..., :CIreal], ref) < 1e-5
    assert float(got[..., CIreal:].abs().sum()) == 0.0


@pytest.mark.parametrize("CI,CIreal,CO,ks,H,W", [(64, 64, 64, 3, 16, 32), (96, 69, 64, 3, 24, 40), (32, 10, 64, 3, 8, 8),
                                                   (64, 64, 60, 1, 12, 36)])
def test_conv_weight_grad_matches_autograd(gpu_device, CI, CIreal, CO, ks, H, W):
    from py4cast_amd import ops_model as om

    g = torch.Generator().manual_seed(11)
    B = 2
    x = torch.randn(B, H, W, CI, generator=g)
    x[..., CIreal:] = 0
    scale = torch.rand(B, CI, generator=g) + 0.5
    shift = torch.randn(B, CI, generator=g) * 0.3
    dout = torch.randn(B, H, W, 64, generator=g)
    dout[..., CO:] = 0
    w = torch.zeros(CO, CIreal, ks, ks, requires_grad=True)
    xin = torch.relu(x * scale[:, None, None, :] + shift[:, None, None, :])
    Fn.conv2d(xin[..., :CIreal].permute(0, 3, 1, 2), w, padding=ks // 2).backward(dout[..., :CO].permute(0, 3, 1, 2))
    grad = torch.ones(CO, CIreal, ks, ks, device=gpu_device)  # accumulation semantics: += on top of ones
    om.conv_wgrad(x.to(gpu_device), dout.to(gpu_device), ks, CO, CIreal, grad, scale.to(gpu_device), shift.to(gpu_device), True)
    assert rel_err(grad - 1.0, w.grad) < 2e-5


def _make_pair(cin, cout, norm, device, seed=0):
    from oracle.halfunet import HalfUNetRef
    from py4cast_amd.halfunet import HalfUNetMI355X, HalfUNetSettings

    torch.manual_seed(seed)
    ref = HalfUNetRef(cin, cout, norm=norm)
    with torch.no_grad():  # non-trivial affine parameters
        for m in ref.modules():
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.GroupNorm)):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    model = HalfUNetMI355X(cin, cout, (32, 32), HalfUNetSettings(norm=norm))
    missing = model.load_state_dict(ref.state_dict(), strict=True)
    return ref, model.to(device)


def _noise_bar(err32, floor):
    """Acceptance bar for a gradient: the larger of `floor` and 10x the error torch's own fp32 CPU path shows
    against the float64 oracle.  ReLU / max-pool arg-max decisions of pre-activations within rounding of a tie
    flip between any two fp32 implementations (torch CPU vs torch GPU vs these kernels), each flip moving a
    gradient by ~1/pixels; these are discrete events with a heavy tail, hence the generous factor.  Wiring or
    indexing mistakes show up as O(0.1..1) errors."""
    return max(floor, 10.0 * err32)


@pytest.mark.parametrize("norm,cin,cout,H,W", [("batch", 69, 60, 64, 64), ("group", 46, 21, 48, 32), ("batch", 10, 1, 64, 96)])
def test_halfunet_forward_backward_match_oracle(gpu_device, norm, cin, cout, H, W):
    import copy

    ref32, model = _make_pair(cin, cout, norm, gpu_device)
    ref64 = copy.deepcopy(ref32).double()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, H, W, cin, generator=g)
    gy = torch.randn(2, H, W, cout, generator=g)

    def run_ref(ref, dt):
        xr = x.detach().clone().to(dt).requires_grad_(True)
        ref.train()
        yr = ref(xr.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        (yr * gy.to(dt)).sum().backward()
        return yr, xr.grad, dict(ref.named_parameters())

    y64, dx64, p64 = run_ref(ref64, torch.float64)
    y32, dx32, p32 = run_ref(ref32, torch.float32)
    xg = x.to(gpu_device).requires_grad_(True)
    model.train()
    yg = model(xg)
    (yg * gy.to(gpu_device)).sum().backward()
    assert yg.shape == y64.shape
    assert rel_err(yg, y64) < 1e-4  # north-star bar for fp32 forward outputs
    nchk = min(cin, 64)
    assert rel_err(xg.grad[..., :nchk], dx64[..., :nchk]) < _noise_bar(rel_err(dx32[..., :nchk], dx64[..., :nchk]), 2e-3)
    for name, p in model.named_parameters():
        assert rel_err(p.grad, p64[name].grad) < _noise_bar(rel_err(p32[name].grad, p64[name].grad), 2e-3), name
    if norm == "batch":  # running statistics follow torch's update rule
        rb = dict(ref64.named_buffers())
        for name, buf in model.named_buffers():
            if buf.dtype.is_floating_point:
                np.testing.assert_allclose(buf.cpu().numpy(), rb[name].float().numpy(), rtol=1e-4, atol=1e-5, err_msg=name)
            else:
                assert int(buf) == int(rb[name])
        ref64.eval(); model.eval()  # eval mode uses the running statistics
        with torch.no_grad():
            ye = model(x.to(gpu_device))
            yre = ref64(x.double().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        assert rel_err(ye, yre) < 1e-4


def test_halfunet_rejects_unsupported_settings():
    from py4cast_amd.halfunet import HalfUNetMI355X, HalfUNetSettings

    with pytest.raises(NotImplementedError):
        HalfUNetMI355X(10, 1, (64, 64), HalfUNetSettings(absolute_pos_embed=True))
    with pytest.raises(NotImplementedError):
        HalfUNetMI355X(10, 1, (64, 64), HalfUNetSettings(num_filters=33))
    with pytest.raises(NotImplementedError):
        HalfUNetMI355X(10, 1, (64, 64), HalfUNetSettings(last_activation="NoSuchActivation"))
    # dilation / num_filters / bias / last_activation / use_ghost are served by the module path (tests/test_round2_gpu.py)
    assert HalfUNetMI355X(10, 1, (64, 64), HalfUNetSettings(dilation=2)).module_path
    assert not HalfUNetMI355X(10, 1, (64, 64), HalfUNetSettings()).module_path


@pytest.mark.parametrize("T", [1, 3])
def test_training_step_with_halfunet_matches_oracle(gpu_device, T):
    """
    AutoRegressiveLightning + HalfUNet on HIP kernels: loss and BPTT gradients vs the float64 CPU oracle.

    T=1 is the strict check.  For T=3 the gradient passes through three randomly initialised BatchNorm/ReLU/
    max-pool networks in sequence and is chaotic in fp32: torch's own CPU and GPU fp32 paths differ from the
    float64 oracle by ~1e-2 there (measured, see DESIGN.md "numerics").  The bar is therefore the larger of a
    floor and 4x the error torch's fp32 CPU run shows on the same quantity; wiring mistakes give O(1) errors.
    """
    import copy

    from helpers import make_batch, make_dataset_info, synthetic_case
    from oracle import losses as olosses
    from oracle import rollout as orollout
    from oracle.halfunet import HalfUNetRef
    from py4cast_amd.lightning import AutoRegressiveLightning

    case = synthetic_case(seed=5, B=2, T=T, H=64, W=64, F=12, Ff=5, Fs=4, border=2)
    info = make_dataset_info(case, 5)
    torch.manual_seed(0)
    lm = AutoRegressiveLightning(
        {}, info, None, num_pred_steps_train=T, batch_size=2, model_name="HalfUNet",
        losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
        training_strategy="scaled_ar",
    )
    ref32 = HalfUNetRef(12 + 4 + 5, 12)
    ref32.load_state_dict(lm.model.state_dict())
    ref64 = copy.deepcopy(ref32).double()
    lm = lm.to(gpu_device)
    lm.train()
    loss = lm.training_step(make_batch(case, gpu_device), 0)
    loss.backward()
    B = 2

    def run_ref(ref, dt):
        c = {k: v.to(dt) for k, v in case.items()}
        statics = c["statics"].unsqueeze(0).expand(B, *c["statics"].shape)
        interior = 1.0 - c["border_mask"]
        ref.train()
        pred = orollout.rollout(ref, c["inputs"], c["forcing"], c["outputs"], statics, c["border_mask"], interior,
                                c["diff_std"], c["diff_mean"], "scaled_ar", 1, False, "train", features_second=True)
        wts = olosses.weighted_loss_weights(c["state_weight"], c["diff_std"], "mse")
        l = olosses.training_loss(pred, c["outputs"], False, [("WeightedLoss", 1.0, dict(weights=wts, interior_mask=interior, kind="mse"))])
        l.backward()
        return l, dict(ref.named_parameters())

    l64, p64 = run_ref(ref64, torch.float64)
    l32, p32 = run_ref(ref32, torch.float32)
    assert abs(loss.item() - l64.item()) / abs(l64.item()) < 1e-4
    floor = 5e-3 if T == 1 else 6e-2
    for name, p in lm.model.named_parameters():
        assert rel_err(p.grad, p64[name].grad) < _noise_bar(rel_err(p32[name].grad, p64[name].grad), floor), name


@pytest.mark.parametrize("strategy,nan,border", [("scaled_ar", False, 2), ("scaled_ar", True, 0), ("diff_ar", False, 0)])
def test_native_rollout_equals_generic_path(gpu_device, strategy, nan, border):
    """The one-node native rollout (K1 -> HalfUNet plan -> fused update+loss, reverse sweep) and the generic per-op
    autograd path run the same kernels: prediction bit-identical, loss and gradients equal to rounding."""
    from helpers import make_batch, make_dataset_info, synthetic_case
    from py4cast_amd.lightning import AutoRegressiveLightning

    case = synthetic_case(seed=9, B=2, T=3, H=32, W=48, F=12, Ff=5, Fs=4, border=border, nan=nan)
    info = make_dataset_info(case, 5)
    torch.manual_seed(0)
    lm = AutoRegressiveLightning(
        {}, info, None, num_pred_steps_train=3, batch_size=2, model_name="HalfUNet",
        losses=[{"class": "WeightedLoss", "weight": 0.7, "params": {"loss": "MSELoss", "reduction": "none"}}],
        training_strategy=strategy, mask_on_nan=nan,
    ).to(gpu_device)
    lm.train()
    res = {}
    # three routes to the same numbers: the one-node native rollout, the generic per-op path with the fused update+loss
    # step (any nn.Module model), and the generic path with separate update and loss passes (the reference's structure)
    for mode, native, fused in (("native", True, True), ("generic_fused", False, True), ("generic", False, False)):
        lm.use_native_rollout, lm.use_fused_step = native, fused
        for p in lm.parameters():
            p.grad = None
        pred, _ = lm.common_step(make_batch(case, gpu_device), 0, "train")
        assert (getattr(pred, "fused_loss", None) is not None) == (native or fused)
        loss = lm.training_step(make_batch(case, gpu_device), 0)
        loss.backward()
        res[mode] = (pred.tensor.detach().cpu(), loss.item(), {n: p.grad.detach().cpu().clone() for n, p in lm.model.named_parameters()})
    for mode in ("native", "generic_fused"):
        a, b = res[mode][0], res["generic"][0]
        assert torch.equal(torch.isnan(a), torch.isnan(b))
        assert rel_err(torch.nan_to_num(a), torch.nan_to_num(b)) < 2e-5  # BatchNorm batch statistics differ in the last bits per call
        assert abs(res[mode][1] - res["generic"][1]) / abs(res["generic"][1]) < 1e-5
        for n in res[mode][2]:
            assert rel_err(res[mode][2][n], res["generic"][2][n]) < 5e-2, (mode, n)  # chaotic BPTT (see test above); typical 1e-4


def test_native_rollout_accumulates_into_flat_grad_buffer(gpu_device):
    """With FlatDDP's layout (every param.grad a view of one flat buffer) the backward plan accumulates straight into
    that buffer: gradients equal the per-parameter path, and a second backward ADDS to them."""
    from helpers import make_batch, make_dataset_info, synthetic_case
    from py4cast_amd.lightning import AutoRegressiveLightning
    from py4cast_amd.trainer import FlatDDP

    case = synthetic_case(seed=4, B=2, T=2, H=32, W=32, F=12, Ff=5, Fs=4, border=0, nan=False)
    info = make_dataset_info(case, 5)
    torch.manual_seed(0)
    lm = AutoRegressiveLightning(
        {}, info, None, num_pred_steps_train=2, batch_size=2, model_name="HalfUNet",
        losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
        training_strategy="scaled_ar",
    ).to(gpu_device)
    lm.train()
    lm.use_native_rollout = True
    for p in lm.parameters():
        p.grad = None
    lm.training_step(make_batch(case, gpu_device), 0).backward()
    ref = {n: p.grad.detach().clone() for n, p in lm.model.named_parameters()}
    ddp = FlatDDP(lm, world_size=1)
    assert lm.model._flat_grad_target() is not None
    for k in (1, 2):
        lm.training_step(make_batch(case, gpu_device), 0).backward()
        for n, p in lm.model.named_parameters():
            assert rel_err(p.grad, k * ref[n]) < 5e-2, (k, n)  # BatchNorm running stats move between calls; typical 1e-4
    assert float(ddp.flat_grad.abs().sum()) > 0


# ----------------------------------------------------------------------------------------------------------------
# bf16 matrix-core flavour: operands are rounded to bf16 (RNE) in LDS, products accumulate in fp32.  The reference
# below applies the same rounding to inputs and weights and convolves in float64, so the only difference left is
# the fp32 accumulation order: the comparison is as strict as the fp32 one and checks every index mapping.
def _bf(t):
    return t.bfloat16().double()


@pytest.mark.parametrize("CI,CIreal,ks,H,W", [(64, 64, 3, 16, 32), (96, 69, 3, 20, 40), (32, 10, 3, 8, 8), (64, 64, 1, 12, 36),
                                                (64, 60, 3, 4, 4), (96, 96, 1, 8, 64), (64, 64, 3, 40, 72)])
def test_conv_fwd_bf16_matches_rounded_reference(gpu_device, CI, CIreal, ks, H, W):
    from py4cast_amd import ops_model as om

    g = torch.Generator().manual_seed(CI + ks + H)
    B, CO = 2, 64
    x = torch.randn(B, H, W, CI, generator=g)
    x[..., CIreal:] = 0
    w = torch.randn(CO, CIreal, ks, ks, generator=g) * 0.1
    scale = torch.rand(B, CI, generator=g) + 0.5
    shift = torch.randn(B, CI, generator=g) * 0.3
    for transform in (False, True):
        xin = torch.relu(x * scale[:, None, None, :] + shift[:, None, None, :]) if transform else x
        ref = Fn.conv2d(_bf(xin[..., :CIreal]).permute(0, 3, 1, 2), _bf(w), padding=ks // 2).permute(0, 2, 3, 1)
        wp = om.prep_weights(w.to(gpu_device), False, 64, CI, compute="bf16")
        out, stats = om.conv_fwd(x.to(gpu_device), wp, ks, in_scale=scale.to(gpu_device) if transform else None,
                                 in_shift=shift.to(gpu_device) if transform else None, in_relu=transform, want_stats=True,
                                 compute="bf16")
        # with the fused input transform an fp32 last-bit difference (FMA contraction) can land on a bf16 rounding
        # boundary and move that operand by one bf16 ulp: ~1e-4 on the output; index mistakes give O(1)
        assert rel_err(out, ref) < (5e-4 if transform else 2e-5)
        s = stats.sum(0).cpu().double()
        np.testing.assert_allclose(s[0].numpy(), ref.sum((0, 1, 2)).numpy(), rtol=2e-3, atol=5e-2)
        np.testing.assert_allclose(s[1].numpy(), (ref**2).sum((0, 1, 2)).numpy(), rtol=2e-3, atol=5e-2)


@pytest.mark.parametrize("CIreal,ks,H,W", [(64, 3, 16, 32), (60, 3, 12, 20), (64, 1, 8, 40)])
def test_conv_data_grad_bf16(gpu_device, CIreal, ks, H, W):
    from py4cast_amd import ops_model as om

    g = torch.Generator().manual_seed(7)
    B, CO = 2, 64
    x = torch.zeros(B, CIreal, H, W, dtype=torch.float64, requires_grad=True)
    w = torch.randn(CO, CIreal, ks, ks, generator=g) * 0.1
    dout = torch.randn(B, H, W, CO, generator=g)
    Fn.conv2d(x, _bf(w), padding=ks // 2).backward(_bf(dout).permute(0, 3, 1, 2))
    ref = x.grad.permute(0, 2, 3, 1)
    wp = om.prep_weights(w.to(gpu_device), True, 64, 64, compute="bf16")
    got = om.conv_fwd(dout.to(gpu_device), wp, ks, compute="bf16")
    assert rel_err(got[..., :CIreal], ref) < 2e-5
    assert float(got[..., CIreal:].abs().sum()) == 0.0


@pytest.mark.parametrize("CI,CIreal,CO,ks,H,W", [(64, 64, 64, 3, 16, 32), (96, 69, 64, 3, 24, 40), (32, 10, 64, 3, 8, 8),
                                                   (64, 64, 60, 1, 12, 36), (64, 64, 64, 3, 40, 72)])
def test_conv_weight_grad_bf16(gpu_device, CI, CIreal, CO, ks, H, W):
    from py4cast_amd import ops_model as om

    g = torch.Generator().manual_seed(11)
    B = 2
    x = torch.randn(B, H, W, CI, generator=g)
    x[..., CIreal:] = 0
    scale = torch.rand(B, CI, generator=g) + 0.5
    shift = torch.randn(B, CI, generator=g) * 0.3
    dout = torch.randn(B, H, W, 64, generator=g)
    dout[..., CO:] = 0
    w = torch.zeros(CO, CIreal, ks, ks, dtype=torch.float64, requires_grad=True)
    xin = torch.relu(x * scale[:, None, None, :] + shift[:, None, None, :])
    Fn.conv2d(_bf(xin[..., :CIreal]).permute(0, 3, 1, 2), w, padding=ks // 2).backward(_bf(dout[..., :CO]).permute(0, 3, 1, 2))
    grad = torch.ones(CO, CIreal, ks, ks, device=gpu_device)
    om.conv_wgrad(x.to(gpu_device), dout.to(gpu_device), ks, CO, CIreal, grad, scale.to(gpu_device), shift.to(gpu_device), True,
                  compute="bf16")
    assert rel_err(grad - 1.0, w.grad) < 5e-4  # fused input transform: see test_conv_fwd_bf16_matches_rounded_reference


def test_halfunet_bf16_close_to_fp32_oracle(gpu_device):
    """End to end with bf16 matrix cores: bf16 operand rounding (2^-9 relative per product) through 13 conv layers."""
    from oracle.halfunet import HalfUNetRef
    from py4cast_amd.halfunet import HalfUNetMI355X, HalfUNetSettings

    torch.manual_seed(0)
    cin, cout, H, W = 69, 60, 64, 64
    ref = HalfUNetRef(cin, cout).double()
    model = HalfUNetMI355X(cin, cout, (H, W), HalfUNetSettings(compute_dtype="bf16", activation_dtype="f32"))
    model.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    model = model.to(gpu_device).train()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, H, W, cin, generator=g)
    gy = torch.randn(2, H, W, cout, generator=g)
    ref.train()
    yr = ref(x.double().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    (yr * gy.double()).sum().backward()
    yg = model(x.to(gpu_device))
    (yg * gy.to(gpu_device)).sum().backward()
    assert rel_err(yg, yr) < 8e-2
    pr = dict(ref.named_parameters())
    for name, p in model.named_parameters():  # direction of the gradient (bf16 noise through 13 ReLU/BN layers is large)
        a, b = p.grad.detach().cpu().double().flatten(), pr[name].grad.flatten()
        assert float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.9, name


@pytest.mark.parametrize("B,H,W", [(3, 202, 72), (9, 24, 40), (33, 8, 32), (2, 512, 64)])
def test_ring_conv_workgroups_walking_strips_and_samples(gpu_device, B, H, W):
    """3x3 64->64 bf16 conv where a persistent workgroup owns SEVERAL tiles: rows inherited through the LDS ring while
    walking down a strip, fresh starts at strip / sample changes, partial last tiles (H % 4, W % 32 != 0), statistics
    flushed per sample; B = 33 exceeds the ring kernel's per-launch sample limit and takes the generic kernel."""
    from py4cast_amd import ops_model as om

    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, H, W, 64, generator=g).bfloat16()
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.1
    scale = torch.rand(B, 64, generator=g) + 0.5
    shift = torch.randn(B, 64, generator=g) * 0.3
    wp = om.prep_weights(w.to(gpu_device), False, 64, 64, compute="bf16")
    for transform in (False, True):
        xin = torch.relu(x.float() * scale[:, None, None, :] + shift[:, None, None, :]) if transform else x.float()
        ref = Fn.conv2d(_bf(xin).permute(0, 3, 1, 2), _bf(w), padding=1).permute(0, 2, 3, 1)
        out, stats = om.conv_fwd(x.to(gpu_device), wp, 3, in_scale=scale.to(gpu_device) if transform else None,
                                 in_shift=shift.to(gpu_device) if transform else None, in_relu=transform, want_stats=True,
                                 compute="bf16")
        assert rel_err(out.float(), ref) < 8e-3
        # per-pixel check catches a misplaced ring row (a whole row of wrong values) that a norm could average away
        err = (out.float().cpu().double() - ref).abs().amax(dim=-1)
        assert float(err.max()) < 0.25, (float(err.max()), torch.nonzero(err > 0.25)[:4])
        o64 = out.cpu().double()
        s = stats.cpu().double().reshape(B, -1, 2, 64).sum(1)   # per-sample statistics
        np.testing.assert_allclose(s[:, 0].numpy(), o64.sum((1, 2)).numpy(), rtol=2e-3, atol=0.5)
        np.testing.assert_allclose(s[:, 1].numpy(), (o64 * o64).sum((1, 2)).numpy(), rtol=1e-2)


@pytest.mark.parametrize("CI,CIreal,ks,H,W", [(64, 64, 3, 16, 32), (96, 69, 3, 20, 40), (32, 10, 3, 8, 8), (64, 64, 1, 12, 36), (64, 64, 3, 40, 72)])
def test_conv_bf16_storage(gpu_device, CI, CIreal, ks, H, W):
    """bf16 activation storage: inputs, outputs and gradients-of-activations live in HBM as bf16 (fp32 accumulate);
    outputs are rounded once to bf16 (2^-9 relative)."""
    from py4cast_amd import ops_model as om

    g = torch.Generator().manual_seed(CI + ks + H)
    B, CO = 2, 64
    x = torch.randn(B, H, W, CI, generator=g).bfloat16()
    x[..., CIreal:] = 0
    w = torch.randn(CO, CIreal, ks, ks, generator=g) * 0.1
    scale = torch.rand(B, CI, generator=g) + 0.5
    shift = torch.randn(B, CI, generator=g) * 0.3
    dout = torch.randn(B, H, W, 64, generator=g).bfloat16()
    for transform in (False, True):
        xin = torch.relu(x.float() * scale[:, None, None, :] + shift[:, None, None, :]) if transform else x.float()
        wd = torch.zeros(CO, CIreal, ks, ks, dtype=torch.float64, requires_grad=True)
        ref = Fn.conv2d(_bf(xin[..., :CIreal]).permute(0, 3, 1, 2), _bf(w) + wd, padding=ks // 2)
        ref.backward(dout.double().permute(0, 3, 1, 2))
        ref = ref.detach().permute(0, 2, 3, 1)
        wp = om.prep_weights(w.to(gpu_device), False, 64, CI, compute="bf16")
        out, stats = om.conv_fwd(x.to(gpu_device), wp, ks, in_scale=scale.to(gpu_device) if transform else None,
                                 in_shift=shift.to(gpu_device) if transform else None, in_relu=transform, want_stats=True,
                                 compute="bf16")
        assert out.dtype == torch.bfloat16
        assert rel_err(out.float(), ref) < 8e-3
        s = stats.sum(0).cpu().double()
        # statistics describe what a consumer will normalise: either the fp32 accumulators (generic kernels) or the
        # bf16-rounded outputs actually stored (3x3 64->64 ring kernel); both are within rounding noise of the reference
        o64 = out.cpu().double()
        own1, own2 = o64.sum((0, 1, 2)).numpy(), (o64 * o64).sum((0, 1, 2)).numpy()
        ref1 = ref.sum((0, 1, 2)).numpy()
        noise = 4 * float(ref.abs().max()) * 2.0**-9 * (B * H * W) ** 0.5
        assert (np.abs(s[0].numpy() - own1) < 1e-3 * np.abs(own1).max() + 1e-3).all() or np.allclose(s[0].numpy(), ref1, rtol=2e-3, atol=5e-2)
        np.testing.assert_allclose(s[0].numpy(), ref1, rtol=2e-3, atol=noise)
        np.testing.assert_allclose(s[1].numpy(), own2, rtol=1e-2)
        grad = torch.zeros(CO, CIreal, ks, ks, device=gpu_device)
        om.conv_wgrad(x.to(gpu_device), dout.to(gpu_device), ks, CO, CIreal, grad, scale.to(gpu_device) if transform else None,
                      shift.to(gpu_device) if transform else None, transform, compute="bf16")
        assert rel_err(grad, wd.grad) < 5e-4


def test_halfunet_bf16_storage_runs_and_tracks_fp32(gpu_device):
    from oracle.halfunet import HalfUNetRef
    from py4cast_amd.halfunet import HalfUNetMI355X, HalfUNetSettings

    torch.manual_seed(0)
    cin, cout, H, W = 69, 60, 64, 64
    ref = HalfUNetRef(cin, cout).double()
    model = HalfUNetMI355X(cin, cout, (H, W), HalfUNetSettings(compute_dtype="bf16", activation_dtype="bf16"))
    model.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    model = model.to(gpu_device).train()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, H, W, cin, generator=g)
    gy = torch.randn(2, H, W, cout, generator=g)
    ref.train()
    yr = ref(x.double().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    (yr * gy.double()).sum().backward()
    yg = model(x.to(gpu_device))
    assert yg.dtype == torch.float32  # the module hands back the caller's dtype
    (yg * gy.to(gpu_device)).sum().backward()
    assert rel_err(yg, yr) < 0.1
    pr = dict(ref.named_parameters())
    for name, p in model.named_parameters():
        a, b = p.grad.detach().cpu().double().flatten(), pr[name].grad.flatten()
        assert float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.85, name


def test_flat_adamw_matches_torch_adamw(gpu_device):
    """FlatAdamW (one kernel over the flat parameter / gradient buffers) against torch.optim.AdamW step by step, with
    the LR changing between steps; state_dict keeps torch's per-parameter layout; scattered parameters fall back."""
    from py4cast_amd.optim import FlatAdamW

    g = torch.Generator().manual_seed(3)
    shapes = [(64, 9, 3, 3), (64,), (64,), (7, 64, 1, 1)]
    n = sum(int(np.prod(s)) for s in shapes)
    flat_p, flat_g = torch.randn(n, generator=g).to(gpu_device), torch.zeros(n, device=gpu_device)
    ps, rs, off = [], [], 0
    for s in shapes:
        k = int(np.prod(s))
        p = torch.nn.Parameter(flat_p[off : off + k].view(s))
        p.grad = flat_g[off : off + k].view(s)
        ps.append(p)
        rs.append(torch.nn.Parameter(p.detach().clone()))
        off += k
    opt = FlatAdamW(ps, lr=1e-3, betas=(0.9, 0.95))
    ref = torch.optim.AdamW(rs, lr=1e-3, betas=(0.9, 0.95))
    for step in range(5):
        gr = torch.randn(n, generator=g).to(gpu_device) * (10.0 ** (step - 2))
        flat_g.copy_(gr)
        off = 0
        for r in rs:
            r.grad = gr[off : off + r.numel()].view_as(r).clone()
            off += r.numel()
        for o in (opt, ref):
            o.param_groups[0]["lr"] = 1e-3 / (step + 1)
            o.step()
        assert opt._flat_state is not None  # the flat path was taken
        for p, r in zip(ps, rs):
            assert rel_err(p.detach(), r.detach()) < 2e-6, step
    sd = opt.state_dict()["state"]
    assert set(sd[0].keys()) >= {"step", "exp_avg", "exp_avg_sq"} and float(sd[0]["step"]) == 5
    for i, r in enumerate(rs):
        assert rel_err(sd[i]["exp_avg_sq"], ref.state[r]["exp_avg_sq"]) < 2e-6
    # not flat -> parent implementation
    q = [torch.nn.Parameter(torch.randn(5, device=gpu_device)), torch.nn.Parameter(torch.randn(7, device=gpu_device))]
    for t in q:
        t.grad = torch.randn_like(t)
    o2 = FlatAdamW(q, lr=1e-2)
    o2.step()
    assert o2._flat_state is None and float(o2.state[q[0]]["step"]) == 1


def test_training_reduces_loss_and_bf16_tracks_fp32(gpu_device):
    """End to end: 25 optimizer steps (native rollout, BPTT, FlatAdamW) on a learnable synthetic task lower the loss in both
    flavours, and the bf16 trajectory stays within a few percent of the fp32 one."""
    from helpers import make_batch, make_dataset_info, synthetic_case
    from py4cast_amd.lightning import AutoRegressiveLightning
    from py4cast_amd.trainer import FlatDDP

    case = synthetic_case(seed=11, B=2, T=2, H=32, W=32, F=12, Ff=5, Fs=4, border=0, nan=False)
    with torch.no_grad():   # next state = a smoothed copy of the previous one: something a conv net can learn
        s, outs = case["inputs"][:, 0], []
        for _ in range(2):
            s = 0.5 * s + 0.5 * torch.roll(s, 1, dims=1)
            outs.append(s)
        case["outputs"] = torch.stack(outs, 1).contiguous()
    info = make_dataset_info(case, 5)
    traj = {}
    for dt in ("f32", "bf16"):
        torch.manual_seed(0)
        lm = AutoRegressiveLightning(
            {"compute_dtype": dt, "activation_dtype": dt}, info, None, num_pred_steps_train=2, batch_size=2, model_name="HalfUNet",
            losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
            training_strategy="scaled_ar", learning_rate=2e-3, num_warmup_steps=0,
        ).to(gpu_device)
        lm.train()
        ddp = FlatDDP(lm.model, 1)
        opt = lm.configure_optimizers()["optimizer"]
        losses = []
        for i in range(25):
            loss = lm.training_step(make_batch(case, gpu_device), i)
            loss.backward()
            opt.step()
            ddp.zero_grad()
            losses.append(float(loss.detach()))
        assert opt._flat_state is not None
        assert losses[-1] < 0.8 * losses[0], losses
        traj[dt] = losses
    for a, b in zip(traj["f32"], traj["bf16"]):
        assert abs(a - b) / abs(a) < 0.08, (traj["f32"], traj["bf16"])


def test_side_stream_switch_reaches_the_backward_thread(gpu_device):
    """p4c_side_stream_enable is process-wide: called from the host's main thread it must reach p4c_halfunet_backward on
    autograd's device thread.  Off: no weight-gradient launch goes to a side stream and the gradients equal the two-stream run's bit for
    bit (the same kernels in the same order per buffer); on again: they go beside the chain."""
    from py4cast_amd import _lib as L

    _, model = _make_pair(69, 60, "batch", gpu_device)
    model.train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 64, 69, generator=g).to(gpu_device)
    gy = torch.randn(2, 64, 64, 60, generator=g).to(gpu_device)
    lib = L.lib()

    def run():
        model.zero_grad(set_to_none=True)
        before = lib.p4c_side_stream_launch_count()
        xg = x.clone().requires_grad_(True)
        (model(xg) * gy).sum().backward()
        torch.cuda.synchronize()
        return lib.p4c_side_stream_launch_count() - before, [p.grad.clone() for p in model.parameters()] + [xg.grad.clone()]

    try:
        L.call("p4c_side_stream_enable", 1)
        n_on, g_on = run()
        L.call("p4c_side_stream_enable", 0)
        n_off, g_off = run()
        L.call("p4c_side_stream_enable", 1)
        n_again, _ = run()
    finally:
        L.call("p4c_side_stream_enable", 1)
    assert n_on > 0 and n_again == n_on
    assert n_off == 0
    for a, b in zip(g_on, g_off):
        assert torch.equal(a, b)
