"""
The network's 1x1 output convolution fused with the AR step (p4c_out_conv_update_loss_fwd, round 4): the north star's "fused
normalise-residual-loss epilogue" of the model's last convolution.  Same arithmetic as the two-kernel route (p4c_conv_fwd 1x1 ->
p4c_ar_update_loss_fwd_next_saved): y is rounded to bf16 where that route stores it, the update runs in the reference's op order
(py4cast/lightning.py:599-633), so the NEW STATE, the next network input and the saved loss gradients are equal bit for bit; the loss
is the same sum in another order.
"""
import ctypes

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("diag_library")]   # (flips P4C_* A/B switches: diagnostic build)

MSE = [{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}]
L1 = [{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "L1Loss", "reduction": "none"}}]


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-30))


@pytest.mark.parametrize("B,H,W,F,Fs,Ff,cpad,kind,border,scaled,nxt,lg", [
    (2, 64, 96, 60, 4, 5, 96, 0, True, True, True, True),        # the benchmark's feature counts
    (1, 8, 125, 60, 4, 5, 96, 1, True, True, True, True),        # N = 1000: not a multiple of 32; L1
    (2, 32, 64, 60, 4, 5, 96, 0, False, True, False, True),      # no border forcing, last AR step (no next input)
    (3, 21, 37, 20, 4, 8, 32, 0, True, False, True, False),      # unscaled update (diff_ar), c_pad 32, no saved gradients
    (2, 64, 64, 32, 0, 0, 32, 1, True, True, True, True),        # nothing but the state in the next input
    # feature counts off the 16-byte grid: the flat AR-step kernel with the convolution as its front end (round 4)
    (2, 64, 80, 21, 4, 21, 64, 0, True, True, True, True),       # the shipped Titan feature counts (C_in = 46 -> 64)
    (1, 8, 125, 21, 4, 21, 64, 1, True, True, True, True),       # N = 1000: a partial last tile; L1
    (2, 27, 76, 5, 3, 2, 32, 0, False, True, False, True),       # odd static / forcing counts, no border forcing, last AR step
    (3, 16, 16, 13, 0, 0, 32, 0, True, False, True, False),      # unscaled update, nothing but the state in the next input
    (2, 32, 64, 62, 1, 1, 64, 0, True, True, True, True),        # rows nearly full: both feature halves of the front end
])
def test_fused_output_conv_and_ar_step_vs_two_kernels(gpu_device, B, H, W, F, Fs, Ff, cpad, kind, border, scaled, nxt, lg):
    from py4cast_amd import _lib as L
    from py4cast_amd import ops_model as om

    dev = gpu_device
    N = H * W
    g = torch.Generator(device=dev).manual_seed(23)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    ru = lambda *s: torch.rand(*s, generator=g, device=dev)
    a = rn(B, H, W, 64).bfloat16()
    sc, sh = ru(B, 64) + 0.5, rn(B, 64) * 0.3
    w = rn(F, 64, 1, 1) * 0.2
    prev, tgt = rn(B, N, F), rn(B, N, F)
    std, mean = (ru(F) + 0.5, rn(F) * 0.01) if scaled else (None, None)
    interior = (ru(N) > 0.2).float()
    bmask = 1.0 - interior if border else None
    weights = ru(F) + 0.5
    statics, forcing = ru(B, N, max(Fs, 1)), ru(B, N, max(Ff, 1))
    ws = torch.empty(L.lib().p4c_loss_workspace_bytes(B, 1, N, 1) // 4, dtype=torch.float32, device=dev)
    st = L.stream(dev)
    num_interior = float(interior.sum())

    def outs():
        return (torch.empty(B, N, F, device=dev), torch.empty(B, device=dev),
                torch.full((B, N, cpad), 7.0, device=dev).bfloat16() if nxt else None,
                torch.full((B, N, F), 7.0, device=dev).bfloat16() if lg else None)

    # two kernels: 1x1 convolution (row kernel: input transform in its loader) -> fused AR step on its bf16 output
    wp = om.prep_weights(w, False, 64, 64, compute="bf16")
    y = om.conv_fwd(a, wp, 1, in_scale=sc, in_shift=sh, in_relu=True, compute="bf16")            # (B,H,W,64) bf16
    ns0, loss0, xn0, lg0 = outs()
    if lg or nxt:
        name = "p4c_ar_update_loss_fwd_next_saved" if lg else "p4c_ar_update_loss_fwd_next"
        args = [L.ptr(prev), N * F, L.ptr(y), L.BF16, 64, L.ptr(tgt), N * F, L.ptr(std), L.ptr(mean), L.ptr(bmask), L.ptr(interior),
                L.ptr(ns0), N * F, L.ptr(weights), num_interior, None, kind, L.MASK_NONE, L.ptr(loss0), 1, L.ptr(ws), B, N, F, 1.0,
                L.ptr(xn0), cpad, L.ptr(statics), N * max(Fs, 1), Fs, L.ptr(forcing), N * max(Ff, 1), Ff]
        if lg:
            args += [L.ptr(lg0), N * F]
        L.call(name, *args, st)
    else:
        L.call("p4c_ar_update_loss_fwd", L.ptr(prev), N * F, L.ptr(y), L.BF16, 64, L.ptr(tgt), N * F, L.ptr(std), L.ptr(mean), L.ptr(bmask),
               L.ptr(interior), L.ptr(ns0), N * F, L.ptr(weights), num_interior, None, kind, L.MASK_NONE, L.ptr(loss0), 1, L.ptr(ws), B, N,
               F, 1.0, st)
    # one kernel
    ns1, loss1, xn1, lg1 = outs()
    wflat = w.reshape(F, 64).contiguous()
    L.call("p4c_out_conv_update_loss_fwd", L.ptr(a), L.ptr(sc), L.ptr(sh), L.ptr(wflat), F, L.ptr(prev), N * F, L.ptr(tgt), N * F,
           L.ptr(std), L.ptr(mean), L.ptr(bmask), L.ptr(interior), L.ptr(ns1), N * F, L.ptr(weights), num_interior, None, kind,
           L.ptr(loss1), 1, L.ptr(ws), B, N, F, 1.0, L.ptr(xn1), cpad, L.ptr(statics), N * max(Fs, 1), Fs, L.ptr(forcing),
           N * max(Ff, 1), Ff, L.ptr(lg1), N * F, st)
    torch.cuda.synchronize()
    assert torch.equal(ns1, ns0), float((ns1 - ns0).abs().max())
    assert rel_err(loss1, loss0) < 1e-6
    if nxt:
        assert torch.equal(xn1.view(torch.int16), xn0.view(torch.int16))
    if lg:
        assert torch.equal(lg1.view(torch.int16), lg0.view(torch.int16))


@pytest.mark.parametrize("losses", [MSE, L1])
@pytest.mark.parametrize("H,W,F,Ff", [(64, 96, 60, 5), (48, 80, 60, 5), (64, 80, 21, 21)])
def test_rollout_with_the_fused_tail_equals_the_two_kernel_route(gpu_device, monkeypatch, losses, H, W, F, Ff):
    """HalfUNet bf16 training step through the native rollout: P4C_FUSED_TAIL=1 (default) vs 0 -- predictions equal bit for bit, the
    loss to summation order, and therefore the parameter gradients bit for bit as well (the saved loss gradients are the same bits);
    also the no-grad (validation) rollout."""
    import bench
    from py4cast_amd.lightning import AutoRegressiveLightning

    case = bench.synthetic_case(91, 2, 3, 1, H, W, F, Ff, 4, 4, gpu_device)
    info = bench.make_info(case, Ff)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("P4C_FUSED_TAIL", mode)
        torch.manual_seed(5)
        lm = AutoRegressiveLightning({"compute_dtype": "bf16", "activation_dtype": "bf16"}, info, None, num_pred_steps_train=3, batch_size=2,
                                     model_name="HalfUNet", losses=losses, training_strategy="scaled_ar").to(gpu_device).train()
        with torch.no_grad():
            pred_ng, _ = lm.common_step(bench.make_batch(case), 0, "train")
        pred_ng = pred_ng.tensor.clone()
        loss = lm.training_step(bench.make_batch(case), 0)
        loss.backward()
        torch.cuda.synchronize()
        res[mode] = (float(loss), torch.cat([q.grad.flatten() for q in lm.model.parameters()]).clone(), pred_ng)
    assert torch.equal(res["1"][2], res["0"][2])
    assert abs(res["1"][0] - res["0"][0]) <= 2e-6 * abs(res["0"][0])
    assert torch.equal(res["1"][1], res["0"][1])
