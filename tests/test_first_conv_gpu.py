"""The first convolution of the bf16 HalfUNet plan at the benchmark's 69 input channels as a 64-channel row launch + a tail pass
(csrc/conv_thin.hip, round 6; replaces the generic K = 96 launch of mfai's first Conv2d, py4cast/lightning.py:591-596):
the tail against float64 on the same bf16 operands, its statistics against the stored output, the whole two-launch convolution
against the one-launch kernel and float64, and the plan with the split against the plan without it (diagnostic switch)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("B,H,W,cin", [(2, 64, 64, 69), (1, 40, 96, 65), (3, 16, 32, 72), (2, 33, 160, 70)])
def test_tail_against_float64(gpu_device, B, H, W, cin):
    """y <- bf16(y1 + conv3x3(x[..., 64:cin])): one rounding of the float64 sum on the same bf16 operands (bf16-rounded weights, as the
    kernel rounds them), image borders, ragged last row groups; channel sums / sums of squares of the STORED result <= 1e-5."""
    from py4cast_amd import _lib as L

    g = torch.Generator().manual_seed(B * 1000 + H + cin)
    x = torch.zeros(B, H, W, 96)
    x[..., :cin] = torch.randn(B, H, W, cin, generator=g)
    w = torch.randn(64, cin, 3, 3, generator=g) * 0.2
    y1 = torch.randn(B, H, W, 64, generator=g)
    xd, yd, wd = _bf(x).to(gpu_device), _bf(y1).to(gpu_device), w.to(gpu_device)
    lib = L.lib()
    slots = lib.p4c_first_conv_tail_slots(B, H, W)
    stats = torch.full((B, slots, 2, 64), float("nan"), device=gpu_device)
    L.call("p4c_first_conv_tail", L.ptr(xd), 96, cin, L.ptr(wd), L.ptr(yd), L.ptr(stats), B, H, W, L.stream(gpu_device))
    ref = _bf(y1).double() + Fn.conv2d(_bf(x)[..., 64:cin].double().permute(0, 3, 1, 2), _bf(w)[:, 64:].double(), padding=1).permute(0, 2, 3, 1)
    got = yd.double().cpu()
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 4e-3, err                       # half a bf16 ulp of the largest value
    # ... and nearly always THE rounding of the float64 sum: fp32 accumulation of 45 exact products moves a value across a rounding
    # boundary only rarely, and then by one bf16 step
    diff = yd.cpu() != _bf(ref)
    assert float(diff.float().mean()) < 2e-3
    step = (yd.cpu().double() - _bf(ref).double()).abs() / ref.abs().clamp_min(1e-3)
    assert float(step.max()) < 1e-2
    s = stats.double().sum(1).cpu()
    np.testing.assert_allclose(s[:, 0].numpy(), got.sum((1, 2)).numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(s[:, 1].numpy(), (got ** 2).sum((1, 2)).numpy(), rtol=1e-5, atol=1e-3)
    # without a statistics buffer: same output
    y2 = _bf(y1).to(gpu_device)
    L.call("p4c_first_conv_tail", L.ptr(xd), 96, cin, L.ptr(wd), L.ptr(y2), None, B, H, W, L.stream(gpu_device))
    assert torch.equal(y2, yd)
    with pytest.raises(L.P4CError):
        L.call("p4c_first_conv_tail", L.ptr(xd), 96, 64, L.ptr(wd), L.ptr(y2), None, B, H, W, L.stream(gpu_device))


def _plan_pair(gpu_device, H, W, seed=0):
    from oracle.halfunet import HalfUNetRef
    from py4cast_amd.halfunet import HalfUNetMI355X, HalfUNetSettings

    torch.manual_seed(seed)
    ref = HalfUNetRef(69, 60).double()
    model = HalfUNetMI355X(69, 60, (H, W), HalfUNetSettings(compute_dtype="bf16", activation_dtype="bf16"))
    model.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    return ref, model.to(gpu_device).train()


def test_plan_with_the_split_first_convolution_tracks_the_one_launch_plan(gpu_device, diag_library, monkeypatch):
    """HalfUNet bf16, 69 -> 60 channels: forward + backward with the first convolution split (default) against the one-launch K = 96
    kernel (P4C_FIRST_CONV_SPLIT=0, diagnostic library) and against the float64 oracle: the split costs one more bf16 rounding of one
    map -- both forms within the bf16 flavour's bar (8e-2) of the oracle and of each other, the split no further from the oracle than the
    one-launch form, gradients aligned, reruns bit-identical."""
    H, W = 64, 96
    ref, model = _plan_pair(gpu_device, H, W)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, H, W, 69, generator=g)
    gy = torch.randn(2, H, W, 60, generator=g)

    def run(split):
        monkeypatch.setenv("P4C_FIRST_CONV_SPLIT", "1" if split else "0")
        model.zero_grad(set_to_none=True)
        xg = x.to(gpu_device).requires_grad_(True)
        y = model(xg)
        (y * gy.to(gpu_device)).sum().backward()
        return y.detach().float().cpu(), xg.grad.float().cpu(), {n: p.grad.float().cpu().clone() for n, p in model.named_parameters()}

    ys, dxs, gs = run(True)
    ys2, dxs2, gs2 = run(True)
    assert torch.equal(ys, ys2) and torch.equal(dxs, dxs2) and all(torch.equal(gs[n], gs2[n]) for n in gs)
    yo, dxo, go = run(False)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())   # noqa: E731
    assert 0 < rel(ys, yo) < 8e-2, rel(ys, yo)          # another rounding of the first map, amplified by 12 more bf16 layers: inside the bf16 bar
    ref.train()
    yr = ref(x.double().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    assert rel(ys, yr) < 8e-2 and rel(yo, yr) < 8e-2
    assert rel(ys, yr) < 1.3 * rel(yo, yr) + 1e-3       # no further from the oracle than the one-launch form (up to noise)
    cos = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm()))   # noqa: E731
    # bf16 noise through 13 ReLU / BN layers: every tensor's gradient points the same way (small vectors -- a 64-entry bias -- are the
    # noisiest), the whole gradient closely so
    for n in gs:
        assert cos(gs[n].double().flatten(), go[n].double().flatten()) > 0.8, n
    ga, gb = torch.cat([gs[n].double().flatten() for n in gs]), torch.cat([go[n].double().flatten() for n in gs])
    assert cos(ga, gb) > 0.9
    assert cos(dxs.double().flatten(), dxo.double().flatten()) > 0.9


def test_split_is_taken_at_the_benchmark_shape_only_where_it_applies(gpu_device):
    """the split needs the bf16 flavour with bf16 activations, 65..72 input channels on 96-channel pixels and a row-kernel grid; the
    Titan configuration (46 -> 64-channel pixels), the fp32 flavour and narrow maps keep their kernels"""
    import ctypes

    from py4cast_amd import _lib as L
    from py4cast_amd.halfunet import HalfUNetMI355X, HalfUNetSettings

    lib = L.lib()
    assert lib.p4c_first_conv_tail_slots(2, 512, 512) == 1024 and lib.p4c_first_conv_tail_slots(2, 16, 32) == 16
    # eval mode / fp32-activation flavour run through without the tail's statistics or without the tail at all
    m = HalfUNetMI355X(69, 60, (64, 64), HalfUNetSettings(compute_dtype="bf16", activation_dtype="bf16")).to(gpu_device)
    x = torch.randn(2, 64, 64, 69, generator=torch.Generator().manual_seed(1)).to(gpu_device)
    m.train()
    yt = m(x)
    m.eval()
    with torch.no_grad():
        ye = m(x)
    assert bool(torch.isfinite(yt).all()) and bool(torch.isfinite(ye).all())
    del ctypes
