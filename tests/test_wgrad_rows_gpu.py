"""
The row-streaming weight-gradient kernel of the 3x3 64 -> 64 convolution (csrc/conv_wgrad_rows.hip, round 4) against float64 on the
SAME bf16-rounded operands (one matmul per tap on the device in float64), against the tile kernel it replaces
(P4C_NO_WGRAD_ROWS=1), for every input transform, with pass 2 of the normalisation backward applied on the way in
(p4c_conv_wgrad_nb), and over segment geometries that exercise every path of its row loop (rows per segment = 0, 1, 2 mod 3; segments
with and without the extra row; first / last strips and segments; one to ten strips).
"""
import os

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("diag_library")]   # (flips P4C_* A/B switches: diagnostic build)


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-30))


def _taps_f64(xin, dy, CO, CI):
    """gw[co][ci][ky][kx] = sum_px xin[px + tap][ci] * dy[px][co] in float64 (xin, dy: (B,H,W,64) float64, "same" zero padding)"""
    B, H, W, C = xin.shape
    xp = torch.nn.functional.pad(xin, (0, 0, 1, 1, 1, 1))
    gw = torch.zeros(C, C, 3, 3, dtype=torch.float64, device=xin.device)
    dyd = dy.reshape(-1, C)
    for ky in range(3):
        for kx in range(3):
            gw[:, :, ky, kx] = dyd.t() @ xp[:, ky:ky + H, kx:kx + W, :].reshape(-1, C)
    return gw[:CO, :CI]


CASES = [
    # B, H, W, forced segments per strip (None: the launcher's choice)
    (2, 64, 64, None),
    (1, 40, 128, None),
    (3, 72, 192, None),
    (2, 27, 64, 1),      # R = 27: 0 mod 3 -> one left-over row
    (2, 28, 64, 1),      # R = 28: two left-over rows
    (2, 29, 64, 1),      # R = 29: none
    (2, 64, 128, 3),     # R = 22 / 21: segments with and without the extra row
    (2, 512, 640, None), # the Titan grid: ten strips, twelve segments of 42 / 43 rows
]


@pytest.mark.parametrize("B,H,W,nseg", CASES)
@pytest.mark.parametrize("mode", ["plain", "relu", "norm_relu", "norm"])
def test_row_streaming_wgrad_vs_float64(gpu_device, monkeypatch, B, H, W, nseg, mode):
    from py4cast_amd import _lib as L
    from py4cast_amd import ops_model as om

    if mode in ("relu", "norm") and (H, W) == (512, 640):
        pytest.skip("the large grid runs the two common modes")
    if nseg is not None:
        monkeypatch.setenv("P4C_WGROWS_NSEG", str(nseg))
    assert L.lib().p4c_conv_wgrad_kernel_kind(L.BF16, B, H, W) == 1
    g = torch.Generator(device=gpu_device).manual_seed(11)
    C, CO, CI = 64, 60, 64
    x = torch.randn(B, H, W, C, generator=g, device=gpu_device).bfloat16()
    dy = torch.randn(B, H, W, C, generator=g, device=gpu_device).bfloat16()
    scale = (torch.rand(B, C, generator=g, device=gpu_device) + 0.5) if mode.startswith("norm") else None
    shift = (torch.randn(B, C, generator=g, device=gpu_device) * 0.3) if mode.startswith("norm") else None
    relu = mode in ("relu", "norm_relu")
    xin = x.float()
    if scale is not None:
        xin = xin * scale[:, None, None, :] + shift[:, None, None, :]
    if relu:
        xin = torch.relu(xin)
    ref = _taps_f64(xin.bfloat16().double(), dy.double(), CO, CI)
    grad = torch.ones(CO, CI, 3, 3, device=gpu_device)           # accumulation semantics: += on top of ones
    om.conv_wgrad(x, dy, 3, CO, CI, grad, scale, shift, relu, compute="bf16")
    err = rel_err(grad - 1.0, ref)
    monkeypatch.setenv("P4C_NO_WGRAD_ROWS", "1")
    assert L.lib().p4c_conv_wgrad_kernel_kind(L.BF16, B, H, W) == 0
    old = torch.ones(CO, CI, 3, 3, device=gpu_device)
    om.conv_wgrad(x, dy, 3, CO, CI, old, scale, shift, relu, compute="bf16")
    assert err < 5e-4, err
    assert rel_err(grad, old) < 5e-5          # same products, fp32 sums in another order


@pytest.mark.parametrize("B,H,W,nseg", [(2, 64, 64, None), (3, 72, 192, None), (2, 28, 128, 1), (2, 256, 256, None)])
@pytest.mark.parametrize("transform", [False, True])
def test_row_streaming_wgrad_with_norm_backward_pass2(gpu_device, monkeypatch, B, H, W, nseg, transform):
    """p4c_conv_wgrad_nb: the gradient operand is dA; dY = alpha * g + beta * y + delta is formed by the staging waves (fp32, one
    rounding to bf16).  Reference: the same formula in float64 on the bf16 inputs, rounded to bf16, then the float64 tap products."""
    from py4cast_amd import ops_model as om

    if nseg is not None:
        monkeypatch.setenv("P4C_WGROWS_NSEG", str(nseg))
    g = torch.Generator(device=gpu_device).manual_seed(13)
    C = 64
    rn = lambda *s: torch.randn(*s, generator=g, device=gpu_device)
    ru = lambda *s: torch.rand(*s, generator=g, device=gpu_device)
    x, dA, y = rn(B, H, W, C).bfloat16(), rn(B, H, W, C).bfloat16(), rn(B, H, W, C).bfloat16()
    gamma, nscale, nshift = ru(C) + 0.5, ru(B, C) + 0.5, rn(B, C) * 0.3
    rstd, mean, k1, k2 = ru(B, C) + 0.5, rn(B, C) * 0.2, rn(B, C) * 0.1, rn(B, C) * 0.1
    scale = (ru(B, C) + 0.5) if transform else None
    shift = (rn(B, C) * 0.3) if transform else None
    bc = lambda t: t[:, None, None, :].double()
    yd, gd = y.double(), dA.double()
    alive = (y.float() * nscale[:, None, None, :] + nshift[:, None, None, :]) > 0     # (fp32 fma in the kernel; ties are measure zero)
    gmask = torch.where(alive, gd, torch.zeros_like(gd))
    al = bc(rstd) * gamma.double()
    be = -(bc(rstd) ** 2) * bc(k2)
    de = bc(rstd) ** 2 * bc(k2) * bc(mean) - bc(rstd) * bc(k1)
    dY = (al * gmask + be * yd + de).float().bfloat16().double()
    xin = x.float()
    if transform:
        xin = torch.relu(xin * scale[:, None, None, :] + shift[:, None, None, :])
    ref = _taps_f64(xin.bfloat16().double(), dY, C, C)
    res = {}
    for off in ("0", "1"):
        monkeypatch.setenv("P4C_NO_WGRAD_ROWS", off)
        grad = torch.zeros(C, C, 3, 3, device=gpu_device)
        om.conv_wgrad_nb(x, dA, y, gamma, nscale, nshift, rstd, mean, k1, k2, C, C, grad, scale, shift, transform)
        res[off] = grad
    # (dY rounded from fp32 in the kernel, from float64 here: a last-bit difference on a few elements)
    assert rel_err(res["0"], ref) < 2e-3, rel_err(res["0"], ref)
    assert rel_err(res["0"], res["1"]) < 5e-5


def test_row_streaming_wgrad_is_reproducible(gpu_device):
    """fixed-order sums: two launches on the same operands give the same bits"""
    from py4cast_amd import ops_model as om

    g = torch.Generator(device=gpu_device).manual_seed(3)
    x = torch.randn(2, 128, 192, 64, generator=g, device=gpu_device).bfloat16()
    dy = torch.randn(2, 128, 192, 64, generator=g, device=gpu_device).bfloat16()
    a = om.conv_wgrad(x, dy, 3, 64, 64, torch.zeros(64, 64, 3, 3, device=gpu_device), compute="bf16")
    b = om.conv_wgrad(x, dy, 3, 64, 64, torch.zeros(64, 64, 3, 3, device=gpu_device), compute="bf16")
    assert torch.equal(a, b)


@pytest.mark.parametrize("B,H,W", [(2, 64, 128), (2, 512, 512)])
@pytest.mark.parametrize("transform", [False, True])
def test_row_streaming_wgrad_of_the_first_convolution(gpu_device, monkeypatch, B, H, W, transform):
    """The 69 -> 64 first convolution (x padded to 96 channels): a full 64-channel chunk and a THIN chunk -- the five real channels
    beyond 64: one octet loaded, half of the matrix waves idle -- into one partial buffer; against float64 and the tile kernels."""
    from py4cast_amd import ops_model as om

    g = torch.Generator(device=gpu_device).manual_seed(17)
    CIp, CI, CO = 96, 69, 64
    x = torch.randn(B, H, W, CIp, generator=g, device=gpu_device)
    x[..., CI:] = 0
    x = x.bfloat16()
    dy = torch.randn(B, H, W, 64, generator=g, device=gpu_device).bfloat16()
    scale = (torch.rand(B, CIp, generator=g, device=gpu_device) + 0.5) if transform else None
    shift = (torch.randn(B, CIp, generator=g, device=gpu_device) * 0.3) if transform else None
    xin = x.float()
    if transform:
        xin = torch.relu(xin * scale[:, None, None, :] + shift[:, None, None, :])
    xin = xin.bfloat16().double()[..., :CI]
    xp = torch.nn.functional.pad(xin, (0, 0, 1, 1, 1, 1))
    ref = torch.zeros(CO, CI, 3, 3, dtype=torch.float64, device=gpu_device)
    dyd = dy.double().reshape(-1, 64)
    for ky in range(3):
        for kx in range(3):
            ref[:, :, ky, kx] = dyd.t() @ xp[:, ky:ky + H, kx:kx + W, :].reshape(-1, CI)
    res = {}
    for off in ("0", "1"):
        monkeypatch.setenv("P4C_NO_WGRAD_ROWS", off)
        grad = torch.zeros(CO, CI, 3, 3, device=gpu_device)
        om.conv_wgrad(x, dy, 3, CO, CI, grad, scale, shift, transform, compute="bf16")
        res[off] = grad
    assert rel_err(res["0"], ref) < 5e-4, rel_err(res["0"], ref)
    assert rel_err(res["0"], res["1"]) < 5e-5
