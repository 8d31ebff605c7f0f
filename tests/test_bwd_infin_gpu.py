"""
Pass 1 of a normalisation backward FINISHED by the kernel that takes it (csrc/norm_pool.hip: bwd_slot_finish, kernels.hpp: BwdFin; round
4): on the coarse levels of the bf16 HalfUNet plan the last workgroup of norm_bwd_reduce / enc_out_bwd sums the launch's slots and
writes d(gamma), d(beta), k1, k2 -- what a norm_bwd_finalize launch did (BatchNorm2d's autograd under py4cast/lightning.py:591-596).
P4C_BWD_INFIN_MAX=0 turns it off (every pass finished by a launch of its own): same sums in another order.
"""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("diag_library")]   # (flips P4C_* A/B switches: diagnostic build)


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-30))


def _model(dev, H, W, norm="batch"):
    from py4cast_amd.halfunet import HalfUNetMI355X, HalfUNetSettings

    torch.manual_seed(3)
    return HalfUNetMI355X(69, 60, (H, W), HalfUNetSettings(norm=norm, compute_dtype="bf16", activation_dtype="bf16")).to(dev)


def _grads(model, x, gy, calls=1):
    for p in model.parameters():
        p.grad = None
    xin = x.clone().requires_grad_(True)
    for _ in range(calls):          # (several backward calls: the ticket must be back at zero after each launch)
        model(xin).backward(gy)
    torch.cuda.synchronize()
    return xin.grad.clone(), {n: p.grad.clone() for n, p in model.named_parameters()}


@pytest.mark.parametrize("shape", [(2, 64, 96), (3, 48, 80), (2, 256, 256), (1, 128, 512)])
def test_eval_mode_gradients_equal_the_separate_finalize(gpu_device, monkeypatch, shape):
    """Eval-mode BatchNorm: k1 = k2 = 0 whatever the sums, so every dA is bit-identical on both routes and only d(gamma) / d(beta)
    -- the sums themselves -- may differ, by fp32 summation order."""
    B, H, W = shape
    model = _model(gpu_device, H, W).eval()
    x = torch.randn(B, H, W, 69, generator=torch.Generator().manual_seed(4)).to(gpu_device)
    gy = torch.randn(B, H, W, 60, generator=torch.Generator().manual_seed(5)).to(gpu_device)
    res = {}
    for mx in ("256", "0"):
        monkeypatch.setenv("P4C_BWD_INFIN_MAX", mx)
        res[mx] = _grads(model, x, gy, calls=3)
    assert torch.equal(res["256"][0], res["0"][0])
    changed = 0
    for n in res["0"][1]:
        e = rel_err(res["256"][1][n], res["0"][1][n])
        if "norm" in n or "bn" in n.lower():
            assert e < 2e-5, (n, e)
            changed += e > 0
        else:
            assert e == 0.0, (n, e)
    # reproducible whichever workgroup comes last
    monkeypatch.setenv("P4C_BWD_INFIN_MAX", "256")
    again = _grads(model, x, gy, calls=3)
    for n in again[1]:
        assert torch.equal(again[1][n], res["256"][1][n]), n


@pytest.mark.parametrize("norm", ["batch", "group"])
def test_train_mode_step_with_the_in_kernel_finish(gpu_device, monkeypatch, norm):
    """Training statistics: k1 / k2 carry the sums, so a changed last bit flips bf16 roundings downstream (see tests/test_round2_gpu.py:
    fused statistics passes) -- direction check; GroupNorm never takes the in-kernel finish (bit-equal)."""
    B, H, W = 2, 64, 96
    model = _model(gpu_device, H, W, norm).train()
    x = torch.randn(B, H, W, 69, generator=torch.Generator().manual_seed(4)).to(gpu_device)
    gy = torch.randn(B, H, W, 60, generator=torch.Generator().manual_seed(5)).to(gpu_device)
    res = {}
    for mx in ("256", "0"):
        monkeypatch.setenv("P4C_BWD_INFIN_MAX", mx)
        res[mx] = _grads(model, x, gy)
    flat = lambda d: torch.cat([v.flatten() for v in d.values()]).double()
    a, b = flat(res["256"][1]), flat(res["0"][1])
    assert torch.isfinite(a).all()
    if norm == "group":
        assert torch.equal(a, b)
    else:
        assert float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.99, rel_err(a, b)


@pytest.mark.parametrize("shape", [(2, 64, 96), (3, 48, 80), (2, 256, 256), (2, 80, 192), (1, 128, 512)])
def test_both_normalisation_backward_roles_in_one_data_gradient_launch(gpu_device, monkeypatch, shape):
    """Round 4: the data-gradient launch of a 64 -> 64 block applies pass 2 of its own normalisation backward in its loader AND takes
    pass 1 of the next block's in its drain (the row kernel at two rows per interval; csrc/conv_rows.hip: IV; P4C_NB_BST=1 -- measured no
    faster, so the plan's default is P4C_NB_BST=0: pass 1 as a norm_bwd_reduce launch over dA and y).  Eval-mode BatchNorm: every dA is bit-identical on both routes, d(gamma) / d(beta) are
    the same sums in another order; train mode: direction check (see tests/test_round2_gpu.py: fused statistics passes)."""
    B, H, W = shape
    x = torch.randn(B, H, W, 69, generator=torch.Generator().manual_seed(4)).to(gpu_device)
    gy = torch.randn(B, H, W, 60, generator=torch.Generator().manual_seed(5)).to(gpu_device)
    model = _model(gpu_device, H, W).eval()
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("P4C_NB_BST", mode)
        res[mode] = _grads(model, x, gy, calls=2)
    assert torch.equal(res["1"][0], res["0"][0])
    for n in res["0"][1]:
        e = rel_err(res["1"][1][n], res["0"][1][n])
        if "norm" in n:
            assert e < 2e-5, (n, e)
        else:
            assert e == 0.0, (n, e)
    model.train()
    for mode in ("1", "0"):
        monkeypatch.setenv("P4C_NB_BST", mode)
        res[mode] = _grads(model, x, gy)
    flat = lambda d: torch.cat([v.flatten() for v in d.values()]).double()
    a, b = flat(res["1"][1]), flat(res["0"][1])
    assert torch.isfinite(a).all()
    assert float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.98, rel_err(a, b)
