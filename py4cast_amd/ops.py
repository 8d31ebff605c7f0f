"""
torch.autograd wrappers around the rollout / loss entry points of ``libpy4cast_hip.so``.

These are thin: they validate layouts, pass raw device pointers + strides + the current
HIP stream through the C ABI and wire the matching backward entry point.  They allocate
outputs with the PyTorch caching allocator (the library owns no memory).
"""

from typing import Optional, Tuple

import torch

from . import _lib as L

_KINDS = {"mse": L.LOSS_MSE, "MSELoss": L.LOSS_MSE, "l1": L.LOSS_L1, "L1Loss": L.LOSS_L1}


def loss_kind_code(kind) -> int:
    try:
        return _KINDS[kind]
    except KeyError:
        raise L.P4CError(f"unsupported torch loss {kind!r}: the HIP losses implement MSELoss and L1Loss") from None


def _rows(t: torch.Tensor, lead: int) -> Tuple[torch.Tensor, list]:
    """
    View ``t`` (lead dims..., *spatial, F) for the kernels: the trailing (*spatial, F) block must
    be dense; leading dims may carry arbitrary strides.  Returns (tensor, leading strides).
    """
    inner = 1
    ok = True
    for d in range(t.dim() - 1, lead - 1, -1):
        if t.size(d) != 1 and t.stride(d) != inner:
            ok = False
            break
        inner *= t.size(d)
    if not ok:
        t = t.contiguous()
    return t, [t.stride(d) if t.size(d) != 1 else 0 for d in range(lead)]


def _numel_spatial(t: torch.Tensor, lead: int) -> int:
    n = 1
    for d in range(lead, t.dim() - 1):
        n *= t.size(d)
    return n


def pad_channels(c: int, multiple: int = 16) -> int:
    return (c + multiple - 1) // multiple * multiple


# ------------------------------------------------------------------------------ K1
class _BuildX(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prev_states, statics, forcing_i, mask_on_nan, downscaling_only, c_pad, out_dtype, blocks=None):
        L.require_cuda(prev_states, statics, forcing_i)
        B, T_in = prev_states.shape[0], prev_states.shape[1]
        F, Fs, Ff = prev_states.shape[-1], statics.shape[-1], forcing_i.shape[-1]
        N = _numel_spatial(forcing_i, 1)
        prev, (pbs, pts) = _rows(prev_states.float(), 2)
        st, (sbs,) = _rows(statics.float(), 1)
        fo, (fbs,) = _rows(forcing_i.float(), 1)
        c_in = (0 if downscaling_only else T_in * F) + Fs + Ff + int(mask_on_nan)
        c_pad = c_in if c_pad is None else c_pad
        x = torch.empty(forcing_i.shape[:-1] + (c_pad,), dtype=out_dtype, device=forcing_i.device)
        args = (L.ptr(prev), pbs, pts, L.ptr(st), sbs, L.ptr(fo), fbs, L.ptr(x), L.dtype_code(out_dtype),
                c_pad, B, T_in, N, F, Fs, Ff, int(mask_on_nan), int(downscaling_only))
        if blocks is None:
            L.call("p4c_build_x", *args, L.stream(x.device))
        else:
            if forcing_i.dim() != 4:
                raise L.P4CError("build_x: the block mask needs the grid layout (B, H, W, features)")
            L.call("p4c_build_x_masked", *args, L.ptr(blocks.selected), forcing_i.shape[1], forcing_i.shape[2], blocks.block_h,
                   blocks.block_w, L.stream(x.device))
        ctx.meta = (B, T_in, N, F, c_pad, bool(downscaling_only), prev_states.shape, blocks, tuple(forcing_i.shape[1:3]))
        return x

    @staticmethod
    def backward(ctx, dx):
        B, T_in, N, F, c_pad, ds, shape, blocks, hw = ctx.meta
        if ds:
            return (torch.zeros(shape, dtype=torch.float32, device=dx.device),) + (None,) * 7
        dx = dx.contiguous()
        dprev = torch.empty(shape, dtype=torch.float32, device=dx.device)
        if blocks is None:
            L.call("p4c_build_x_bwd", L.ptr(dx), L.dtype_code(dx.dtype), c_pad, L.ptr(dprev), B, T_in, N, F, L.stream(dx.device))
        else:
            L.call("p4c_build_x_bwd_masked", L.ptr(dx), L.dtype_code(dx.dtype), c_pad, L.ptr(dprev), B, T_in, N, F,
                   L.ptr(blocks.selected), hw[0], hw[1], blocks.block_h, blocks.block_w, L.stream(dx.device))
        return (dprev,) + (None,) * 7


class BlockMask:
    """The draw of ``mask_tensor`` (lightning.py:769-785) as a table: ``selected`` (H*W,) uint8 on the device, 1 where the flat
    index was drawn; grid point (y, x) is cleared iff ``selected[(y // block_h) * W + x // block_w]``."""

    def __init__(self, selected: torch.Tensor, block_h: int, block_w: int):
        self.selected, self.block_h, self.block_w = selected, int(block_h), int(block_w)

    @staticmethod
    def draw(height: int, width: int, mask_ratio: float, device) -> "BlockMask":
        """Same arithmetic and the same torch CPU generator draw as the reference, so the same mask bit for bit."""
        num_blocks = int((1 - mask_ratio) * height * width)
        block_h = height // int(height**0.5)
        block_w = width // int(width**0.5)
        drawn = torch.randperm(height * width)[:num_blocks]
        selected = torch.zeros(height * width, dtype=torch.uint8)
        selected[drawn] = 1
        return BlockMask(selected.to(device), block_h, block_w)

    def dense(self, height: int, width: int) -> torch.Tensor:
        """(H, W) bool, True = kept (the reference's ``mask[0, :, :, 0]``); for observers and tests."""
        dev = self.selected.device
        by = torch.arange(height, device=dev) // self.block_h
        bx = torch.arange(width, device=dev) // self.block_w
        return self.selected[by[:, None] * width + bx[None, :]] == 0


def build_x(prev_states, statics, forcing_i, mask_on_nan=False, downscaling_only=False, c_pad=None, dtype=torch.float32,
            blocks: Optional[BlockMask] = None):
    """lightning.py:711-767.  prev_states (B,T_in,*S,F); statics (B,*S,Fs); forcing_i (B,*S,Ff) -> (B,*S,c_pad).
    ``blocks``: the masked-auto-encoder block mask (lightning.py:580-581, 769-785) applied in the same pass."""
    return _BuildX.apply(prev_states, statics, forcing_i, mask_on_nan, downscaling_only, c_pad, dtype, blocks)


# ------------------------------------------------------------------------------ K2
class _ARUpdate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prev, y, border_state, std, mean, border_mask, interior_mask, keep_prev, nan_to_num):
        L.require_cuda(y)
        B = y.shape[0]
        if prev is not None:
            F = prev.shape[-1]
        elif border_state is not None:
            F = border_state.shape[-1]
        else:
            F = std.numel() if std is not None else y.shape[-1]
        y_c = y.contiguous()
        y_cs = y_c.shape[-1]
        N = _numel_spatial(y_c, 1)
        pv, pbs = (None, 0)
        if prev is not None:
            pv, (pbs,) = _rows(prev.float(), 1)
        bsv, bbs = (None, 0)
        force = border_mask is not None
        if force:
            bsv, (bbs,) = _rows(border_state.float(), 1)
        new_state = torch.empty(y_c.shape[:-1] + (F,), dtype=torch.float32, device=y.device)
        L.call(
            "p4c_ar_update_fwd", L.ptr(pv), pbs, L.ptr(y_c), L.dtype_code(y_c.dtype), y_cs, L.ptr(bsv), bbs, L.ptr(std),
            L.ptr(mean), L.ptr(border_mask if force else None), L.ptr(interior_mask if force else None),
            L.ptr(new_state), N * F, B, N, F, float(keep_prev), int(nan_to_num), L.stream(y.device),
        )
        ctx.save_for_backward(std, interior_mask if force else None)
        ctx.meta = (B, N, F, y_cs, float(keep_prev), y.dtype, prev is not None, prev.shape if prev is not None else None)
        return new_state

    @staticmethod
    def backward(ctx, dnew):
        std, interior = ctx.saved_tensors
        B, N, F, y_cs, keep, ydt, has_prev, pshape = ctx.meta
        dnew, (dbs,) = _rows(dnew.float(), 1)
        dy = torch.empty(dnew.shape[:-1] + (y_cs,), dtype=ydt, device=dnew.device)
        dprev = torch.empty(pshape, dtype=torch.float32, device=dnew.device) if has_prev else None
        L.call(
            "p4c_ar_update_bwd", L.ptr(dnew), dbs, L.ptr(std), L.ptr(interior), L.ptr(dy), L.dtype_code(ydt), y_cs,
            L.ptr(dprev), N * F, B, N, F, keep, L.stream(dnew.device),
        )
        return dprev, dy, None, None, None, None, None, None, None


def ar_update(prev, y, border_state=None, std=None, mean=None, border_mask=None, interior_mask=None, keep_prev=1.0,
              nan_to_num=False):
    """lightning.py:599-633 in one pass.  masks are flat (N,) float tensors; returns fp32 (B,*S,F)."""
    return _ARUpdate.apply(prev, y, border_state, std, mean, border_mask, interior_mask, keep_prev, nan_to_num)


# ------------------------------------------------------------------------------ K3
class MaskSpec:
    """How the loss mask is provided (see p4c_mask_mode in include/py4cast_hip.h)."""

    def __init__(self, mode: int, tensor: Optional[torch.Tensor] = None):
        self.mode, self.tensor = mode, tensor

    @staticmethod
    def from_tensor(mask: Optional[torch.Tensor]) -> "MaskSpec":
        if mask is None:
            return MaskSpec(L.MASK_NONE)
        if mask.dtype == torch.bool:
            return MaskSpec(L.MASK_U8, mask.contiguous().view(torch.uint8))
        if mask.dtype == torch.uint8:
            return MaskSpec(L.MASK_U8, mask.contiguous())
        return MaskSpec(L.MASK_F32, mask.float().contiguous())


def masked_count(spec: MaskSpec, target: torch.Tensor) -> Optional[torch.Tensor]:
    """#grid points masked for every (b,t,f) -- the correction of losses.py:156,167.  None when mask==1."""
    if spec.mode == L.MASK_NONE:
        return None
    B, T, F = target.shape[0], target.shape[1], target.shape[-1]
    N = _numel_spatial(target, 2)
    count = torch.empty(1, dtype=torch.int32, device=target.device)
    if spec.mode == L.MASK_FROM_NAN:
        src, (bs, ts) = _rows(target, 2)
    else:
        src, bs, ts = spec.tensor, T * N * F, N * F
    L.call("p4c_mask_all_zero_count", L.ptr(src), spec.mode, bs, ts, B, T, N, F, L.ptr(count), L.stream(target.device))
    return count


def _workspace(B, T, N, F, device):
    nbytes = L.lib().p4c_loss_workspace_bytes(B, T, N, F)
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device)


class _WeightedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, spec, weights, interior, num_interior, count, kind):
        L.require_cuda(pred, target)
        B, T, F = pred.shape[0], pred.shape[1], pred.shape[-1]
        N = _numel_spatial(pred, 2)
        p, (pbs, pts) = _rows(pred.float(), 2)
        g, (gbs, gts) = _rows(target.float(), 2)
        out = torch.empty(B, T, dtype=torch.float32, device=pred.device)
        ws = _workspace(B, T, N, F, pred.device)
        L.call(
            "p4c_weighted_loss_fwd", L.ptr(p), pbs, pts, L.ptr(g), gbs, gts, L.ptr(spec.tensor), spec.mode,
            L.ptr(weights), L.ptr(interior), float(num_interior), L.ptr(count), kind, L.ptr(out), L.ptr(ws), B, T, N, F,
            L.stream(pred.device),
        )
        ctx.save_for_backward(p, g, weights, interior, count, spec.tensor)
        ctx.meta = (spec.mode, float(num_interior), kind, pred.shape)
        return out

    @staticmethod
    def backward(ctx, gout):
        p, g, weights, interior, count, mtensor = ctx.saved_tensors
        mode, num_interior, kind, shape = ctx.meta
        B, T, F = shape[0], shape[1], shape[-1]
        N = _numel_spatial(p, 2)
        _, (pbs, pts) = _rows(p, 2)
        _, (gbs, gts) = _rows(g, 2)
        dpred = torch.empty(shape, dtype=torch.float32, device=p.device)
        L.call(
            "p4c_weighted_loss_bwd", L.ptr(gout.contiguous().float()), L.ptr(p), pbs, pts, L.ptr(g), gbs, gts,
            L.ptr(mtensor), mode, L.ptr(weights), L.ptr(interior), num_interior, L.ptr(count), kind, L.ptr(dpred),
            T * N * F, N * F, B, T, N, F, L.stream(p.device),
        )
        return dpred, None, None, None, None, None, None, None


def weighted_loss(pred, target, spec: MaskSpec, weights, interior, num_interior, kind, count=None):
    """losses.py:130-169 with reduce_spatial_dim=True -> (B,T)."""
    if count is None:
        count = masked_count(spec, target)
    return _WeightedLoss.apply(pred, target, spec, weights, interior, num_interior, count, kind)


def weighted_loss_map(pred, target, spec: MaskSpec, weights, kind):
    """losses.py:144-154 with reduce_spatial_dim=False -> (B,T,*S).  Not differentiable (plots only)."""
    L.require_cuda(pred, target)
    B, T, F = pred.shape[0], pred.shape[1], pred.shape[-1]
    N = _numel_spatial(pred, 2)
    p, (pbs, pts) = _rows(pred.detach().float(), 2)
    g, (gbs, gts) = _rows(target.detach().float(), 2)
    out = torch.empty(pred.shape[:-1], dtype=torch.float32, device=pred.device)
    L.call(
        "p4c_weighted_loss_map", L.ptr(p), pbs, pts, L.ptr(g), gbs, gts, L.ptr(spec.tensor), spec.mode, L.ptr(weights),
        kind, L.ptr(out), B, T, N, F, L.stream(pred.device),
    )
    return out


def scaled_loss(pred, target, spec: MaskSpec, std, interior, num_interior, kind, count=None):
    """losses.py:186-210 -> (B,T,F).  Metric path (val/test): not differentiable."""
    L.require_cuda(pred, target)
    B, T, F = pred.shape[0], pred.shape[1], pred.shape[-1]
    N = _numel_spatial(pred, 2)
    if count is None:
        count = masked_count(spec, target)
    p, (pbs, pts) = _rows(pred.detach().float(), 2)
    g, (gbs, gts) = _rows(target.detach().float(), 2)
    out = torch.empty(B, T, F, dtype=torch.float32, device=pred.device)
    ws = _workspace(B, T, N, F, pred.device)
    L.call(
        "p4c_scaled_loss_fwd", L.ptr(p), pbs, pts, L.ptr(g), gbs, gts, L.ptr(spec.tensor), spec.mode, L.ptr(std),
        L.ptr(interior), float(num_interior), L.ptr(count), kind, L.ptr(out), L.ptr(ws), B, T, N, F, L.stream(pred.device),
    )
    return out


# ------------------------------------------------------------------------------ K2+K3 fused training step
class _ARStepLoss(torch.autograd.Function):
    """
    One AR step of the training path: state update + that step's loss column in one pass.
    Inputs: prev (B,*S,F) or None, y (B,*S,y_cs) from the model, target (B,*S,F) = outputs[:, i].
    Outputs: new_state (B,*S,F) and loss (B,).
    """

    @staticmethod
    def forward(ctx, prev, y, target, std, mean, border_mask, interior_mask, weights, num_interior, count,
                kind, mask_mode, keep_prev, force_border, out=None):
        L.require_cuda(y, target)
        B, F = target.shape[0], target.shape[-1]
        y_c = y.contiguous()
        y_cs = y_c.shape[-1]
        N = _numel_spatial(target, 1)
        pv, pbs = (None, 0)
        if prev is not None:
            pv, (pbs,) = _rows(prev, 1)
        tg, (tbs,) = _rows(target, 1)
        if out is not None:   # caller-provided slot of the (B,T,*S,F) prediction buffer: no stack/copy afterwards
            assert out.shape == target.shape and out.dtype == torch.float32 and out[0].is_contiguous()
            new_state, nbs = out, out.stride(0)
        else:
            new_state, nbs = torch.empty(target.shape, dtype=torch.float32, device=y.device), N * F
        ns = new_state
        loss = torch.empty(B, dtype=torch.float32, device=y.device)
        ws = _workspace(B, 1, N, 1, y.device)
        L.call(
            "p4c_ar_update_loss_fwd", L.ptr(pv), pbs, L.ptr(y_c), L.dtype_code(y_c.dtype), y_cs, L.ptr(tg), tbs,
            L.ptr(std), L.ptr(mean), L.ptr(border_mask if force_border else None), L.ptr(interior_mask), L.ptr(ns), nbs,
            L.ptr(weights), float(num_interior), L.ptr(count), kind, mask_mode, L.ptr(loss), 1, L.ptr(ws), B, N, F,
            float(keep_prev), L.stream(y.device),
        )
        ctx.save_for_backward(ns, tg, std, interior_mask, weights, count)
        ctx.meta = (B, N, F, y_cs, y.dtype, nbs, tbs, float(num_interior), kind, mask_mode, float(keep_prev),
                    bool(force_border), prev is not None, prev.shape if prev is not None else None)
        return new_state, loss

    @staticmethod
    def backward(ctx, g_new, g_loss):
        ns, tg, std, interior, weights, count = ctx.saved_tensors
        B, N, F, y_cs, ydt, nbs, tbs, num_interior, kind, mask_mode, keep, force, has_prev, pshape = ctx.meta
        gn, gnbs = (None, 0)
        if g_new is not None:
            gn, (gnbs,) = _rows(g_new.float(), 1)
        gl = g_loss.contiguous().float() if g_loss is not None else None
        dy = torch.empty(tg.shape[:-1] + (y_cs,), dtype=ydt, device=tg.device)
        dprev = torch.empty(pshape, dtype=torch.float32, device=tg.device) if has_prev else None
        L.call(
            "p4c_ar_update_loss_bwd", L.ptr(gn), gnbs, None, L.dtype_code(ydt), 0, L.ptr(gl), 1, L.ptr(ns), nbs,
            L.ptr(tg), tbs, L.ptr(std), L.ptr(interior), int(force), L.ptr(weights), num_interior, L.ptr(count), kind,
            mask_mode, L.ptr(dy), L.dtype_code(ydt), y_cs, L.ptr(dprev), N * F, B, N, F, keep, L.stream(tg.device),
        )
        return (dprev, dy) + (None,) * 13


def ar_step_loss(prev, y, target, std, mean, border_mask, interior_mask, weights, num_interior, count, kind,
                 mask_mode, keep_prev, force_border, out=None):
    return _ARStepLoss.apply(prev, y, target, std, mean, border_mask, interior_mask, weights, num_interior,
                             count, kind, mask_mode, keep_prev, force_border, out)


# ------------------------------------------------------------------------------------------------ rows next to the path
def unnormalize(x: torch.Tensor, std: torch.Tensor, mean: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """out = x*std + mean per feature (last dim), as the reference's two in-place passes (lightning.py:1162-1169)."""
    L.require_cuda(x)
    x = x.contiguous().float()
    F = x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    L.call("p4c_unnormalize", L.ptr(x), L.ptr(std.to(x).contiguous()), L.ptr(mean.to(x).contiguous()), L.ptr(out),
           x.numel() // F, F, L.stream(x.device))
    return out


def unnormalize_planes(x: torch.Tensor, std: torch.Tensor, mean: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """x (B,T,*S,F) -> (B,T,F,*S): un-normalised, one contiguous plane per (sample, time step, feature) -- the layout the GRIB / GIF
    writers consume (io/outputs.py:116-241), same two rounded steps as ``unnormalize``."""
    L.require_cuda(x)
    x = x.contiguous().float()
    B, T, F = x.shape[0], x.shape[1], x.shape[-1]
    spatial = tuple(x.shape[2:-1])
    N = 1
    for d in spatial:
        N *= d
    if out is None:
        out = torch.empty((B, T, F) + spatial, dtype=torch.float32, device=x.device)
    L.call("p4c_unnormalize_planes", L.ptr(x), L.ptr(std.to(x).contiguous()), L.ptr(mean.to(x).contiguous()), L.ptr(out), B * T, N, F,
           L.stream(x.device))
    return out


def acc_sums(pred: torch.Tensor, target: torch.Tensor, spec: "MaskSpec", climate_means: torch.Tensor) -> torch.Tensor:
    """Spatial means of (p-c)(t-c)m, ((p-c)m)^2, ((t-c)m)^2 -> (3,B,T,F) (MetricACC.update, metrics.py:414-423)."""
    L.require_cuda(pred, target)
    p, g = pred.contiguous().float(), target.contiguous().float()
    B, T, F = p.shape[0], p.shape[1], p.shape[-1]
    N = p.numel() // (B * T * F)
    out = torch.empty(3, B, T, F, dtype=torch.float32, device=p.device)
    ws = torch.empty(3 * L.lib().p4c_loss_workspace_bytes(B, T, N, F) // 4, dtype=torch.float32, device=p.device)
    L.call("p4c_acc_sums", L.ptr(p), T * N * F, N * F, L.ptr(g), T * N * F, N * F, L.ptr(spec.tensor), spec.mode,
           L.ptr(climate_means.to(p).contiguous()), L.ptr(out), L.ptr(ws), B, T, N, F, L.stream(p.device))
    return out


def pack_standardize(raw: torch.Tensor, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """raw (F, *lead) parameter planes -> (*lead, F) standardised features-last tensor (base.py:448-452 + concat)."""
    L.require_cuda(raw)
    raw = raw.contiguous().float()
    F = raw.shape[0]
    rows = raw.numel() // F
    out = torch.empty(*raw.shape[1:], F, dtype=torch.float32, device=raw.device)
    L.call("p4c_pack_standardize", L.ptr(raw), rows, L.ptr(mean.to(raw).contiguous()), L.ptr(std.to(raw).contiguous()),
           L.ptr(out), rows, F, L.stream(raw.device))
    return out
