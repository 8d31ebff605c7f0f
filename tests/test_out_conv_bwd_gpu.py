"""
The fused backward of the network's 1x1 output convolution (csrc/out_conv_bwd.hip, p4c_out_conv_bwd; round 4): data gradient, pass 1
of the last block's normalisation backward and the weight gradient in one pass over dy and y -- what autograd does for mfai's
`outconv` under py4cast/lightning.py:591-596.  Against
  * the three kernels it replaces in the backward plan (1x1 data gradient on the row kernel: BIT-equal dA; 1x1 weight gradient);
  * float64 on the same bf16 operands (statistics, weight gradient);
  * the whole HalfUNet training step with the fusion on / off (P4C_FUSED_OUT_BWD).
"""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("diag_library")]   # (flips P4C_* A/B switches: diagnostic build)

MSE = [{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}]


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-30))


@pytest.mark.parametrize("B,H,W,CO", [
    (2, 64, 96, 60),        # the benchmark's feature count
    (1, 8, 125, 60),        # N = 1000: a partial last tile
    (3, 8, 8, 21),          # one tile per sample
    (2, 40, 72, 64),        # every output channel real
    (5, 16, 48, 7),         # more samples than the usual batch, few channels
    (2, 512, 640, 21),      # the shipped Titan grid and feature count
])
def test_fused_output_conv_backward_vs_three_kernels_and_float64(gpu_device, B, H, W, CO):
    from py4cast_amd import _lib as L
    from py4cast_amd import ops_model as om

    dev = gpu_device
    N = H * W
    g = torch.Generator(device=dev).manual_seed(31)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    ru = lambda *s: torch.rand(*s, generator=g, device=dev)
    dy = rn(B, N, 64)
    dy[..., CO:] = 0
    dy = dy.bfloat16()
    y = rn(B, N, 64).bfloat16()
    w = rn(CO, 64) * 0.2
    scale, shift, mean, rstd = ru(B, 64) + 0.5, rn(B, 64) * 0.3, rn(B, 64) * 0.2, ru(B, 64) + 0.5

    gw = torch.ones(CO, 64, device=dev)                       # accumulation semantics: += on top of ones
    dA, stats = om.out_conv_bwd(dy, w, y, scale, shift, mean, rstd, gw)
    torch.cuda.synchronize()
    assert stats.shape[1] == L.lib().p4c_out_conv_bwd_slots(B, N)

    # data gradient: the row kernel's 1x1 launch on the same prepared weights -- the same MFMA chain per element
    wpt = om.prep_weights(w.reshape(CO, 64, 1, 1), True, 64, 64, compute="bf16")
    dA_rows = om.conv_fwd(dy.view(B, H, W, 64), wpt, 1, compute="bf16").view(B, N, 64)
    assert torch.equal(dA.view(torch.int16), dA_rows.view(torch.int16))
    dA64 = dy.double()[..., :CO] @ w.bfloat16().double()         # dA[px][ci] = sum_co dy[px][co] * W[co][ci]
    assert rel_err(dA, dA64) < 4e-3                             # (one rounding to bf16)

    # statistics: sums of g and g * xhat over each sample, g = dA where the forward ReLU was alive
    bc = lambda t: t[:, None, :].double()
    alive = (y.double() * bc(scale) + bc(shift)) > 0
    gm = torch.where(alive, dA.double(), torch.zeros((), dtype=torch.float64, device=dev))
    s1 = gm.sum(1)
    s2 = (gm * (y.double() - bc(mean)) * bc(rstd)).sum(1)
    got = stats.double().sum(1)                                 # (B, 2, 64)
    assert rel_err(got[:, 0], s1) < 2e-5, rel_err(got[:, 0], s1)
    assert rel_err(got[:, 1], s2) < 2e-5, rel_err(got[:, 1], s2)

    # weight gradient: float64 on the bf16-rounded operands, and the 1x1 weight-gradient kernel it replaces
    a_n = torch.relu((y.double() * bc(scale) + bc(shift)).float().bfloat16().double())
    ref = torch.einsum("bnc,bnk->ck", dy.double()[..., :CO], a_n)
    assert rel_err(gw - 1.0, ref) < 5e-4, rel_err(gw - 1.0, ref)
    old = torch.ones(CO, 64, 1, 1, device=dev)
    om.conv_wgrad(y.view(B, H, W, 64), dy.view(B, H, W, 64), 1, CO, 64, old, scale, shift, True, compute="bf16")
    assert rel_err(gw, old.view(CO, 64)) < 5e-5

    # fixed-order sums: a second launch gives the same bits
    gw2 = torch.ones(CO, 64, device=dev)
    dA2, stats2 = om.out_conv_bwd(dy, w, y, scale, shift, mean, rstd, gw2)
    assert torch.equal(gw2, gw) and torch.equal(stats2, stats) and torch.equal(dA2.view(torch.int16), dA.view(torch.int16))


@pytest.mark.parametrize("H,W,F", [(64, 96, 60), (48, 80, 21)])
def test_training_step_with_the_fused_output_conv_backward(gpu_device, monkeypatch, H, W, F):
    """HalfUNet bf16 training step through the native rollout with the fusion on (default) and off: the same loss (the forward does
    not change); train-mode gradients are a direction check only -- a changed last bit of a statistic flips bf16 roundings and, on
    these small grids, max-pool arg-maxes downstream (tests/test_round2_gpu.py: fused statistics passes); the exact statement is the
    eval-mode test below."""
    import bench
    from py4cast_amd.lightning import AutoRegressiveLightning

    case = bench.synthetic_case(91, 2, 3, 1, H, W, F, 5, 4, 4, gpu_device)
    info = bench.make_info(case, 5)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("P4C_FUSED_OUT_BWD", mode)
        torch.manual_seed(5)
        lm = AutoRegressiveLightning({"compute_dtype": "bf16", "activation_dtype": "bf16"}, info, None, num_pred_steps_train=3, batch_size=2,
                                     model_name="HalfUNet", losses=MSE, training_strategy="scaled_ar").to(gpu_device).train()
        loss = lm.training_step(bench.make_batch(case), 0)
        loss.backward()
        torch.cuda.synchronize()
        res[mode] = (float(loss.detach()), {n: q.grad.clone() for n, q in lm.model.named_parameters()})
    assert res["1"][0] == res["0"][0]
    flat = lambda d: torch.cat([v.flatten() for v in d.values()]).double()
    a, b = flat(res["1"][1]), flat(res["0"][1])
    assert torch.isfinite(a).all()
    assert float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.99, rel_err(a, b)


@pytest.mark.parametrize("shape", [(2, 64, 96), (3, 48, 80), (2, 256, 256)])
def test_eval_mode_gradients_with_the_fused_output_conv_backward(gpu_device, monkeypatch, shape):
    """Eval-mode BatchNorm: dY = scale * g, so every layer's dA is bit-identical with and without the fusion and the gradients differ by
    the order of fp32 sums only: the output convolution's weight gradient and the last block's d(gamma) / d(beta) (the sums the fused
    kernel forms); everything upstream of them equal bit for bit."""
    from py4cast_amd.halfunet import HalfUNetMI355X, HalfUNetSettings

    B, H, W = shape
    torch.manual_seed(3)
    model = HalfUNetMI355X(69, 60, (H, W), HalfUNetSettings(norm="batch", compute_dtype="bf16", activation_dtype="bf16")).to(gpu_device).eval()
    x = torch.randn(B, H, W, 69, generator=torch.Generator().manual_seed(4)).to(gpu_device)
    gy = torch.randn(B, H, W, 60, generator=torch.Generator().manual_seed(5)).to(gpu_device)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("P4C_FUSED_OUT_BWD", mode)
        for p in model.parameters():
            p.grad = None
        xin = x.clone().requires_grad_(True)
        model(xin).backward(gy)
        torch.cuda.synchronize()
        res[mode] = (xin.grad.clone(), {n: p.grad.clone() for n, p in model.named_parameters()})
    assert torch.equal(res["1"][0], res["0"][0])
    errs = {n: rel_err(res["1"][1][n], res["0"][1][n]) for n in res["0"][1]}
    print({k: v for k, v in errs.items() if v > 0})
    assert sum(v > 0 for v in errs.values()) <= 3, errs          # the output weight, the last block's gamma and beta
    assert max(errs.values()) < 5e-5, errs
