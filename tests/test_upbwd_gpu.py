"""
x pass of the adjoint of HalfUNet's decoder merge (p4c_upsample_sum_bwd_x: csrc/upbwd_mfma.hip on the matrix cores for bf16 rows with
W % 64 == 0, csrc/norm_pool.hip on the vector ALU otherwise) -- the backward of mfai's `F.interpolate(level_k, scale_factor=2^k,
mode="bilinear")` sum under py4cast/lightning.py:591-596.  Against float64 with the interpolation matrix taken from torch itself
(F.interpolate of an identity), on the same operands; the two kernels against each other; bit-identical reruns.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("diag_library")]   # (P4C_UPBWD_VALU selects the older kernel: diagnostic build)


def interp_matrix(Wk, s):
    """(W, Wk) float64: column X = the bilinear (align_corners=False) up-sampling of the unit vector e_X to W = s * Wk points."""
    eye = torch.eye(Wk, dtype=torch.float64).reshape(Wk, 1, 1, Wk)
    up = F.interpolate(eye.expand(Wk, 1, 2, Wk), scale_factor=(1, s), mode="bilinear", align_corners=False)   # (Wk,1,2,W)
    return up[:, 0, 0, :].t().contiguous()


def run(dS, storage):
    from py4cast_amd import _lib as L

    B, H, W, _ = dS.shape
    outs = [torch.full((B, H, W >> k, 64), float("nan"), device=dS.device, dtype=dS.dtype) for k in (1, 2, 3, 4)]
    L.call("p4c_upsample_sum_bwd_x", storage, L.ptr(dS), B, H, W, *[L.ptr(o) for o in outs], L.stream())
    return outs


@pytest.mark.parametrize("B,H,W", [(2, 16, 64), (1, 5, 128), (2, 19, 512), (3, 7, 192), (1, 9, 48), (2, 3, 16)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_x_pass_of_the_upsample_adjoint_against_float64(gpu_device, monkeypatch, B, H, W, dtype):
    from py4cast_amd import _lib as L

    g = torch.Generator(device=gpu_device).manual_seed(7)
    dS = torch.randn(B, H, W, 64, generator=g, device=gpu_device).to(dtype)
    storage = L.BF16 if dtype == torch.bfloat16 else L.F32
    outs = run(dS, storage)
    ref_in = dS.double().cpu()
    for k, got in zip((1, 2, 3, 4), outs):
        M = interp_matrix(W >> k, 1 << k)                       # (W, Wk)
        ref = torch.einsum("bhxc,xX->bhXc", ref_in, M)
        assert torch.isfinite(got).all()
        tol = 2.0 ** -8 if dtype == torch.bfloat16 else 1e-6    # one bf16 rounding of the result
        err = (got.double().cpu() - ref).abs().max() / ref.abs().max()
        assert float(err) <= tol, (k, float(err))
    again = run(dS, storage)
    for a, b in zip(outs, again):
        assert torch.equal(a, b)
    if dtype == torch.bfloat16 and W % 64 == 0:
        monkeypatch.setenv("P4C_UPBWD_VALU", "1")
        old = run(dS, storage)
        for k, (a, b) in enumerate(zip(outs, old)):
            # same exact products, another order of the fp32 sum: at most the last bf16 digit
            assert float((a.float() - b.float()).abs().max()) <= 2.0 ** -7 * float(b.float().abs().max()), k
