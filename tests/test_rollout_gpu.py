"""
GPU parity of the rollout / loss HIP kernels (through the C ABI) against the CPU oracle and
against the golden vectors of the unmodified reference.  Bars: bit-exact for index/mask ops
and for the elementwise state update; <=1e-5 relative for reductions (tolerance target of
the north star: 1e-4).
"""
import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden
from helpers import (TinyConvModel, TinyLinearModel, make_batch, make_dataset_info, register_test_models,
                     synthetic_case)
from oracle import losses as olosses
from oracle import rollout as orollout

pytestmark = pytest.mark.gpu
FILES = golden_files()


def _to(case, dev):
    return {k: v.to(dev) for k, v in case.items()}


@pytest.mark.parametrize("nan", [False, True])
@pytest.mark.parametrize("T_in,F,Ff,Fs", [(1, 60, 5, 4), (2, 5, 7, 4), (1, 21, 21, 4), (3, 70, 3, 5)])
def test_build_x_bit_exact(gpu_device, nan, T_in, F, Ff, Fs):
    from py4cast_amd import ops

    case = synthetic_case(seed=3, H=24, W=20, T_in=T_in, F=F, Ff=Ff, Fs=Fs, nan=nan)
    B = case["inputs"].shape[0]
    statics = case["statics"].unsqueeze(0).expand(B, *case["statics"].shape)
    forcing_i = case["forcing"][:, 1]
    ref = orollout.next_x(case["inputs"], statics, forcing_i, T_in, mask_on_nan=nan)
    d = _to(case, gpu_device)
    got = ops.build_x(d["inputs"], d["statics"].unsqueeze(0).expand(B, *case["statics"].shape), d["forcing"][:, 1], nan)
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got.cpu().numpy(), ref.float().numpy())
    # padded + bf16 variant: extra channels are zero, values are the bf16 rounding of the fp32 ones
    c_pad = ops.pad_channels(ref.shape[-1])
    gotp = ops.build_x(d["inputs"], d["statics"].unsqueeze(0).expand(B, *case["statics"].shape), d["forcing"][:, 1], nan,
                       c_pad=c_pad, dtype=torch.bfloat16)
    np.testing.assert_array_equal(gotp[..., : ref.shape[-1]].float().cpu().numpy(), ref.float().bfloat16().float().numpy())
    assert float(gotp[..., ref.shape[-1]:].abs().sum()) == 0.0


def test_build_x_backward(gpu_device):
    from py4cast_amd import ops

    case = _to(synthetic_case(seed=4, H=8, W=12, T_in=2, F=5), gpu_device)
    prev = case["inputs"].clone().requires_grad_(True)
    B = prev.shape[0]
    st = case["statics"].unsqueeze(0).expand(B, *case["statics"].shape)
    x = ops.build_x(prev, st, case["forcing"][:, 0], c_pad=32)
    g = torch.randn_like(x)
    x.backward(g)
    ref = torch.stack([g[..., :5], g[..., 5:10]], dim=1)
    np.testing.assert_array_equal(prev.grad.cpu().numpy(), ref.cpu().numpy())


@pytest.mark.parametrize("scaled,border,nan", [(True, 2, False), (True, 0, True), (False, 0, False), (True, 3, True)])
def test_ar_update_bit_exact(gpu_device, scaled, border, nan):
    from py4cast_amd import ops

    case = synthetic_case(seed=5, H=16, W=24, F=60, border=border, nan=nan)
    g = torch.Generator().manual_seed(9)
    y = torch.randn(case["outputs"][:, 0].shape, generator=g)
    prev = case["inputs"][:, -1].clone()
    bs = case["outputs"][:, 0].clone()
    bm, im = case["border_mask"], 1.0 - case["border_mask"]
    lp = torch.nan_to_num(prev, nan=0) if nan else prev
    if scaled:
        pred = lp * 1 + y * case["diff_std"] + case["diff_mean"]
    else:
        pred = lp * 1 + y
    force = scaled
    ref = bm * (torch.nan_to_num(bs, nan=0) if nan else bs) + im * pred if force else pred
    d = _to(case, gpu_device)
    got = ops.ar_update(
        d["inputs"][:, -1], y.to(gpu_device), d["outputs"][:, 0] if force else None,
        d["diff_std"] if scaled else None, d["diff_mean"] if scaled else None,
        bm.reshape(-1).to(gpu_device) if force else None, im.reshape(-1).to(gpu_device) if force else None,
        keep_prev=1.0, nan_to_num=nan,
    )
    np.testing.assert_array_equal(got.cpu().numpy(), ref.numpy())  # same op order, no FMA contraction


def _loss_inputs(seed, nan, border, F=60, H=24, W=20):
    case = synthetic_case(seed=seed, H=H, W=W, F=F, border=border, nan=nan)
    g = torch.Generator().manual_seed(seed + 100)
    pred = torch.randn(case["outputs"].shape, generator=g)
    return case, pred


@pytest.mark.parametrize("kind", ["mse", "l1"])
@pytest.mark.parametrize("nan,border,F", [(False, 0, 60), (True, 2, 60), (True, 0, 5), (False, 3, 21), (False, 1, 130)])
def test_losses_match_oracle(gpu_device, kind, nan, border, F):
    from py4cast_amd import _lib as L
    from py4cast_amd import ops

    case, pred = _loss_inputs(11, nan, border, F=F)
    interior = 1.0 - case["border_mask"]
    mask, tgt = orollout.get_mask_on_nan(case["outputs"], nan)
    wts = olosses.weighted_loss_weights(case["state_weight"], case["diff_std"], kind)
    ref_w = olosses.weighted_loss(pred, tgt, mask, wts, interior, kind)
    ref_map = olosses.weighted_loss(pred, tgt, mask, wts, interior, kind, reduce_spatial_dim=False)
    ref_s = olosses.scaled_loss(pred, tgt, mask, case["std"], interior, kind)
    dev = gpu_device
    code = ops.loss_kind_code(kind)
    interior_flat = interior.reshape(-1).to(dev)
    num_interior = float(interior.sum())
    specs = {"from_nan": (ops.MaskSpec(L.MASK_FROM_NAN), case["outputs"].to(dev))} if nan else {"none": (ops.MaskSpec(L.MASK_NONE), tgt.to(dev))}
    specs["f32"] = (ops.MaskSpec.from_tensor(mask.float().to(dev)), tgt.to(dev))
    specs["bool"] = (ops.MaskSpec.from_tensor((mask != 0).to(dev)), tgt.to(dev))
    for name, (spec, target) in specs.items():
        got_w = ops.weighted_loss(pred.to(dev), target, spec, wts.to(dev), interior_flat, num_interior, code)
        got_map = ops.weighted_loss_map(pred.to(dev), target, spec, wts.to(dev), code)
        got_s = ops.scaled_loss(pred.to(dev), target, spec, case["std"].to(dev), interior_flat, num_interior, code)
        np.testing.assert_allclose(got_w.cpu().numpy(), ref_w.numpy(), rtol=1e-5, err_msg=name)
        np.testing.assert_allclose(got_map.cpu().numpy(), ref_map.numpy(), rtol=1e-5, atol=1e-6, err_msg=name)
        np.testing.assert_allclose(got_s.cpu().numpy(), ref_s.numpy(), rtol=1e-5, err_msg=name)


@pytest.mark.parametrize("kind", ["mse", "l1"])
@pytest.mark.parametrize("nan,border", [(False, 0), (True, 2)])
def test_weighted_loss_backward(gpu_device, kind, nan, border):
    from py4cast_amd import _lib as L
    from py4cast_amd import ops

    case, pred = _loss_inputs(12, nan, border)
    interior = 1.0 - case["border_mask"]
    mask, tgt = orollout.get_mask_on_nan(case["outputs"], nan)
    wts = olosses.weighted_loss_weights(case["state_weight"], case["diff_std"], kind)
    p = pred.clone().requires_grad_(True)
    gsel = torch.randn(pred.shape[:2], generator=torch.Generator().manual_seed(1))
    (olosses.weighted_loss(p, tgt, mask, wts, interior, kind) * gsel).sum().backward()
    dev = gpu_device
    pg = pred.to(dev).requires_grad_(True)
    spec = ops.MaskSpec(L.MASK_FROM_NAN if nan else L.MASK_NONE)
    out = ops.weighted_loss(pg, case["outputs"].to(dev), spec, wts.to(dev), interior.reshape(-1).to(dev), float(interior.sum()),
                            ops.loss_kind_code(kind))
    (out * gsel.to(dev)).sum().backward()
    np.testing.assert_allclose(pg.grad.cpu().numpy(), p.grad.numpy(), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("scaled,border,nan,kind", [(True, 2, False, "mse"), (True, 2, True, "mse"), (False, 0, False, "l1")])
def test_fused_step_equals_unfused(gpu_device, scaled, border, nan, kind):
    """K2+K3 fused kernel == K2 followed by K3 (values bit-exact for the state, 1e-6 for the loss; grads too)."""
    from py4cast_amd import _lib as L
    from py4cast_amd import ops

    dev = gpu_device
    case = synthetic_case(seed=21, H=16, W=24, F=60, border=border, nan=nan)
    d = _to(case, dev)
    y0 = torch.randn(case["outputs"][:, 0].shape, generator=torch.Generator().manual_seed(2)).to(dev)
    interior = (1.0 - d["border_mask"]).reshape(-1).contiguous()
    bmask = d["border_mask"].reshape(-1).contiguous()
    wts = olosses.weighted_loss_weights(case["state_weight"], case["diff_std"], kind).to(dev)
    num_interior = float(interior.sum())
    code = ops.loss_kind_code(kind)
    mode = L.MASK_FROM_NAN if nan else L.MASK_NONE
    spec = ops.MaskSpec(mode)
    tgt_raw = d["outputs"][:, 0:1]
    count = ops.masked_count(spec, tgt_raw)
    std, mean = (d["diff_std"], d["diff_mean"]) if scaled else (None, None)
    res = {}
    for variant in ("unfused", "fused"):
        prev = d["inputs"][:, -1].clone().requires_grad_(True)
        y = y0.clone().requires_grad_(True)
        if variant == "unfused":
            new = ops.ar_update(prev, y, tgt_raw[:, 0] if scaled else None, std, mean, bmask if scaled else None,
                                interior if scaled else None, 1.0, nan)
            loss = ops.weighted_loss(new.unsqueeze(1), tgt_raw, spec, wts, interior, num_interior, code, count=count)[:, 0]
        else:
            new, loss = ops.ar_step_loss(prev, y, tgt_raw[:, 0], std, mean, bmask, interior, wts, num_interior, count,
                                         code, mode, 1.0, scaled)
        gn = torch.randn(new.shape, generator=torch.Generator().manual_seed(3)).to(dev)
        (loss.sum() * 0.7 + (new * gn).sum()).backward()
        res[variant] = (new.detach().cpu().numpy(), loss.detach().cpu().numpy(), prev.grad.cpu().numpy(), y.grad.cpu().numpy())
    np.testing.assert_array_equal(res["fused"][0], res["unfused"][0])
    np.testing.assert_allclose(res["fused"][1], res["unfused"][1], rtol=1e-6)
    np.testing.assert_allclose(res["fused"][2], res["unfused"][2], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(res["fused"][3], res["unfused"][3], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("path", FILES, ids=lambda p: p.split("/")[-1][:-4])
def test_common_step_matches_reference_golden(gpu_device, path):
    """AutoRegressiveLightning._common_step + losses + BPTT on HIP kernels vs the unmodified reference's outputs."""
    from py4cast_amd.lightning import AutoRegressiveLightning

    register_test_models()
    meta, ins, outs = load_golden(path)
    case = {k: torch.from_numpy(v) for k, v in ins.items()}
    Ff = case["forcing"].shape[-1]
    info = make_dataset_info(case, Ff)
    grid = meta["layout"] == "grid"
    lm = AutoRegressiveLightning(
        {}, info, None, num_input_steps=meta["T_in"], num_pred_steps_train=3, batch_size=2,
        model_name="TinyConvModel" if grid else "TinyLinearModel",
        losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
        num_inter_steps=meta["K"], training_strategy=meta["strategy"], mask_on_nan=bool(meta["nan"]),
    )
    with torch.no_grad():
        lm.model.w.copy_(case["w"])
        lm.model.b.copy_(case["b"])
    lm = lm.to(gpu_device)
    batch = make_batch(case, gpu_device)
    pred, tgt = lm._common_step(batch, 0, "train")
    got = pred.tensor.detach().cpu().numpy()
    ref = outs["prediction"]
    assert got.shape == ref.shape
    np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
    # the conv/tanh of the tiny model run on different hardware (MIOpen vs CPU): 1e-5 abs on O(1) states
    np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(ref), rtol=1e-4, atol=2e-5)
    if not grid:
        return
    mask, tgt_masked = lm.get_mask_on_nan(tgt)
    from py4cast_amd.losses import ScaledLoss, WeightedLoss

    for tag, cls, kind in [("wmse", WeightedLoss, "MSELoss"), ("wl1", WeightedLoss, "L1Loss"), ("smse", ScaledLoss, "MSELoss"), ("sl1", ScaledLoss, "L1Loss")]:
        lossobj = cls(kind, reduction="none")
        lossobj.prepare(lm, lm.interior_mask, info)
        val = lossobj(pred, tgt_masked, mask)
        np.testing.assert_allclose(val.detach().cpu().numpy(), outs[f"loss_{tag}"], rtol=2e-4, atol=1e-6, err_msg=tag)
        if cls is WeightedLoss:
            vmap = lossobj(pred, tgt_masked, mask, reduce_spatial_dim=False)
            np.testing.assert_allclose(vmap.detach().cpu().numpy(), outs[f"loss_{tag}_map"], rtol=2e-3, atol=2e-4, err_msg=tag)
    batch2 = make_batch(case, gpu_device)
    loss = lm.training_step(batch2, 0)
    loss.backward()
    np.testing.assert_allclose(loss.item(), outs["train_loss"], rtol=1e-4)
    np.testing.assert_allclose(lm.model.w.grad.cpu().numpy(), outs["grad_w"], rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(lm.model.b.grad.cpu().numpy(), outs["grad_b"], rtol=2e-3, atol=2e-5)
    # validation_step (lightning.py:888-917): mean over batch and steps of the same loss, with and without the fused step
    for fused in (True, False):
        lm.use_fused_step = fused
        val = lm.validation_step(make_batch(case, gpu_device), 0)
        np.testing.assert_allclose(float(val), float(np.mean(outs["loss_wmse"])), rtol=2e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("F,Ff,Fs,cpad,border", [(60, 5, 4, 96, 0), (12, 7, 4, 24, 2), (20, 4, 8, 32, 0)])
def test_fused_step_emits_next_input_bit_exact(gpu_device, dtype, F, Ff, Fs, cpad, border):
    """p4c_ar_update_loss_fwd_next: same new state and loss as p4c_ar_update_loss_fwd, and its x_next is bit-identical to
    p4c_build_x applied to that new state ("feed next step", lightning.py:636-656 + 711-767)."""
    import ctypes

    from py4cast_amd import _lib as L
    from py4cast_amd import ops

    g = torch.Generator().manual_seed(F * 7 + Ff)
    B, H, W, T = 2, 24, 20, 2
    N = H * W
    dev = gpu_device
    prev = torch.randn(B, H, W, F, generator=g).to(dev)
    y = (torch.randn(B, H, W, 64, generator=g)).to(dev).to(dtype)
    target = torch.randn(B, H, W, F, generator=g).to(dev)
    statics = torch.rand(B, H, W, Fs, generator=g).to(dev)
    forcing_next = (torch.rand(B, H, W, Ff, generator=g) * 100).to(dev)
    std, mean = (torch.rand(F, generator=g) + 0.5).to(dev), (torch.randn(F, generator=g) * 0.01).to(dev)
    weights = (torch.rand(F, generator=g) + 0.5).to(dev)
    bm = torch.zeros(H, W)
    if border:
        bm[:border] = 1; bm[-border:] = 1; bm[:, :border] = 1; bm[:, -border:] = 1
    border_flat, interior_flat = bm.reshape(-1).to(dev), (1 - bm).reshape(-1).to(dev)
    num_interior = float(interior_flat.sum())
    acode = L.dtype_code(dtype)
    ws = torch.empty(L.lib().p4c_loss_workspace_bytes(B, 1, N, 1) // 4, dtype=torch.float32, device=dev)
    stream = L.stream(dev)
    outs = {}
    for fused in (False, True):
        new = torch.empty(B, H, W, F, device=dev)
        loss = torch.empty(B, device=dev)
        args = (L.ptr(prev), N * F, L.ptr(y), acode, 64, L.ptr(target), N * F, L.ptr(std), L.ptr(mean),
                L.ptr(border_flat if border else None), L.ptr(interior_flat), L.ptr(new), N * F, L.ptr(weights), num_interior,
                None, L.LOSS_MSE, L.MASK_NONE, L.ptr(loss), 1, L.ptr(ws), B, N, F, 1.0)
        xn = None
        if fused:
            xn = torch.full((B, H, W, cpad), float("nan"), device=dev).to(dtype)
            L.call("p4c_ar_update_loss_fwd_next", *args, L.ptr(xn), cpad, L.ptr(statics), N * Fs, Fs, L.ptr(forcing_next), N * Ff, Ff, stream)
        else:
            L.call("p4c_ar_update_loss_fwd", *args, stream)
        outs[fused] = (new, loss, xn)
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    ref_x = ops.build_x(outs[False][0][:, None], statics, forcing_next, c_pad=cpad, dtype=dtype)
    assert torch.equal(outs[True][2], ref_x)
