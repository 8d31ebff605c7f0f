"""Thin wrappers over the single-op model entry points (p4c_conv_fwd / p4c_conv_wgrad / p4c_prep_weights):
used by the parity tests and available for composing other conv models behind the plugin API."""

import os
import threading
from typing import Optional

import torch

from . import _lib as L


def _compute(compute) -> int:
    return {"f32": L.F32, "bf16": L.BF16, L.F32: L.F32, L.BF16: L.BF16}[compute]


def prep_weights(w: torch.Tensor, transpose_flip: bool, M_pad: int, K_pad: int, compute="f32") -> torch.Tensor:
    """w: canonical (CO,CI,ks,ks) fp32 -> MFMA operand stream (see include/py4cast_hip.h)."""
    CO, CI, ks, _ = w.shape
    dt = torch.float32 if _compute(compute) == L.F32 else torch.bfloat16
    out = torch.empty(M_pad * K_pad * ks * ks, dtype=dt, device=w.device)
    L.call("p4c_prep_weights", L.ptr(w.contiguous()), CO, CI, ks, int(transpose_flip), M_pad, K_pad, L.ptr(out),
           _compute(compute), L.stream(w.device))
    return out


def conv_tiles(B, H, W, CI=64, compute="f32", storage=None, ks=3):
    storage = _compute(compute) if storage is None else storage
    return B * L.lib().p4c_conv_stat_tiles_ks(_compute(compute), storage, CI, ks, B, H, W)


def conv_fwd(x: torch.Tensor, wprep: torch.Tensor, ks: int, m_blocks: int = 1, in_scale: Optional[torch.Tensor] = None,
             in_shift: Optional[torch.Tensor] = None, in_relu: bool = False, bias: Optional[torch.Tensor] = None,
             want_stats: bool = False, compute="f32"):
    """x: (B,H,W,CI) fp32 (or bf16 with compute="bf16") NHWC, CI in {32,64,96} -> (B,H,W,64*m_blocks) of x's dtype
    [+ per-tile channel sums]."""
    L.require_cuda(x)
    B, H, W, CI = x.shape
    out = torch.empty(B, H, W, 64 * m_blocks, dtype=x.dtype, device=x.device)
    stats = torch.empty(conv_tiles(B, H, W, CI, compute, storage=L.dtype_code(x.dtype), ks=ks), 2, 64, dtype=torch.float32, device=x.device) if want_stats else None
    L.call("p4c_conv_fwd", L.ptr(x.contiguous()), _compute(compute), L.dtype_code(x.dtype), CI, L.ptr(wprep), ks, L.ptr(in_scale), L.ptr(in_shift), int(in_relu),
           L.ptr(bias), L.ptr(out), 64 * m_blocks, L.ptr(stats), B, H, W, m_blocks, L.stream(x.device))
    return (out, stats) if want_stats else out


def conv_wgrad(x: torch.Tensor, dout: torch.Tensor, ks: int, CO: int, CI: int, grad: torch.Tensor,
               in_scale: Optional[torch.Tensor] = None, in_shift: Optional[torch.Tensor] = None, in_relu: bool = False,
               compute="f32"):
    """grad (CO,CI,ks,ks) += sum_px act(x)[px+tap][ci] * dout[px][co];  x (B,H,W,CI_pad), dout (B,H,W,64)."""
    L.require_cuda(x, dout, grad)
    B, H, W, CIp = x.shape
    nbytes = L.lib().p4c_conv_wgrad_workspace_bytes(CIp, ks)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    L.call("p4c_conv_wgrad", L.ptr(x.contiguous()), _compute(compute), L.dtype_code(x.dtype), CIp, ks, L.ptr(in_scale), L.ptr(in_shift), int(in_relu),
           L.ptr(dout.contiguous()), CO, CI, L.ptr(grad), L.ptr(ws), B, H, W, L.stream(x.device))
    return grad


def conv_wgrad_nb(x: torch.Tensor, dA: torch.Tensor, y: torch.Tensor, gamma, nscale, nshift, rstd, mean, k1, k2, CO: int, CI: int,
                  grad: torch.Tensor, in_scale: Optional[torch.Tensor] = None, in_shift: Optional[torch.Tensor] = None, in_relu: bool = False):
    """grad (CO,CI,3,3) += weight gradient of a 3x3 64 -> 64 convolution whose gradient operand is dA of its [norm -> ReLU]: pass 2 of the
    normalisation backward is applied on the way in (p4c_conv_wgrad_nb; bf16 maps)."""
    L.require_cuda(x, dA, y, grad)
    B, H, W, _ = x.shape
    ws = torch.empty(L.lib().p4c_conv_wgrad_workspace_bytes(64, 3) // 4, dtype=torch.float32, device=x.device)
    L.call("p4c_conv_wgrad_nb", L.ptr(x.contiguous()), L.ptr(in_scale), L.ptr(in_shift), int(in_relu), L.ptr(dA.contiguous()), L.ptr(y.contiguous()),
           L.ptr(gamma), L.ptr(nscale), L.ptr(nshift), L.ptr(rstd), L.ptr(mean), L.ptr(k1), L.ptr(k2), CO, CI, L.ptr(grad), L.ptr(ws), B, H, W,
           L.stream(x.device))
    return grad


def out_conv_bwd(dy: torch.Tensor, w: torch.Tensor, y: torch.Tensor, scale, shift, mean, rstd, grad_w: torch.Tensor):
    """The whole backward of a 1x1 convolution from 64 channels behind [conv -> norm -> ReLU] in one pass (p4c_out_conv_bwd; bf16 maps):
    dy (B,N,64) with channels >= CO zero, w (CO,64) fp32, y (B,N,64) the block's raw output, scale / shift / mean / rstd (B,64).
    Returns dA (B,N,64) bf16 and the statistics slots (B, slots, 2, 64) of pass 1 of the normalisation backward; grad_w (CO,64) +=."""
    L.require_cuda(dy, w, y, grad_w)
    B, N, _ = dy.shape
    CO = w.shape[0]
    wp = prep_weights(w.detach().float().reshape(CO, 64, 1, 1), True, 64, 64, compute="bf16")
    slots = L.lib().p4c_out_conv_bwd_slots(B, N)
    dA = torch.empty(B, N, 64, dtype=torch.bfloat16, device=dy.device)
    stats = torch.empty(B, slots, 2, 64, dtype=torch.float32, device=dy.device)
    ws = torch.empty(L.lib().p4c_out_conv_bwd_workspace_bytes(B, N) // 4, dtype=torch.float32, device=dy.device)
    L.call("p4c_out_conv_bwd", L.ptr(dy.contiguous()), L.ptr(wp), L.ptr(y.contiguous()), L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(rstd),
           L.ptr(dA), L.ptr(stats), CO, L.ptr(grad_w), L.ptr(ws), B, N, L.stream(dy.device))
    return dA, stats


# ------------------------------------------------------------------------------ a whole convolution as one autograd node
def _pad32(c: int) -> int:
    return (c + 31) // 32 * 32


class _ConvNHWC(torch.autograd.Function):
    """"same" ks x ks convolution (ks = 1 | 3, stride 1, no bias) to 64 output channels on features-last tensors, all three passes
    on the native kernels: forward p4c_conv_fwd, data gradient the same kernel on the transposed / flipped weights, weight
    gradient p4c_conv_wgrad.  x (B,H,W,Cp) with Cp = CI rounded up to 32 (extra channels zero), of the activation dtype."""

    @staticmethod
    def forward(ctx, x, w, compute):
        L.require_cuda(x, w)
        CO, CI, ks, _ = w.shape
        wp = prep_weights(w.detach().float(), False, 64, x.shape[-1], compute=compute)
        ctx.save_for_backward(x, w)
        ctx.compute = compute
        return conv_fwd(x, wp, ks, compute=compute)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        CO, CI, ks, _ = w.shape
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            mb = (x.shape[-1] + 63) // 64
            wpt = prep_weights(w.detach().float(), True, 64 * mb, 64, compute=ctx.compute)
            dx = conv_fwd(dy, wpt, ks, m_blocks=mb, compute=ctx.compute)[..., : x.shape[-1]]
            if dx.shape[-1] != x.shape[-1] or not dx.is_contiguous():
                dx = dx.contiguous()
        if ctx.needs_input_grad[1]:
            dw = torch.zeros(CO, CI, ks, ks, dtype=torch.float32, device=x.device)
            conv_wgrad(x, dy, ks, CO, CI, dw, compute=ctx.compute)
            dw = dw.to(w.dtype)
        return dx, dw, None


class _ConvCompact(torch.autograd.Function):
    """The same convolution on bf16 feature maps with fewer than 64 channels, IN PLACE (p4c_conv_fwd_compact / p4c_conv_wgrad_compact):
    x (B,H,W,C) with C a multiple of 8 up to 64, CO likewise -- no zero-padded 64-channel copy of x, no sliced 64-channel result, and in
    the backward no zero-filled 64-channel gradient: the row kernel stages absent channel octets as zeros and stores only the present
    ones.  (SwinUNetR's decoder at 24 / 48 channels: these copies were most of its 552 layout-copy launches per training step.)"""

    @staticmethod
    def forward(ctx, x, w):
        L.require_cuda(x, w)
        CO, CI, ks, _ = w.shape
        B, H, W, C = x.shape
        wp = prep_weights(w.detach().float(), False, 64, 64, compute="bf16")
        out = torch.empty(B, H, W, CO, dtype=x.dtype, device=x.device)
        L.call("p4c_conv_fwd_compact", L.ptr(x), C, L.ptr(wp), ks, L.ptr(out), CO, B, H, W, L.stream(x.device))
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        CO, CI, ks, _ = w.shape
        B, H, W, C = x.shape
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wpt = prep_weights(w.detach().float(), True, 64, 64, compute="bf16")
            dx = torch.empty_like(x)
            L.call("p4c_conv_fwd_compact", L.ptr(dy), CO, L.ptr(wpt), ks, L.ptr(dx), C, B, H, W, L.stream(x.device))
        if ctx.needs_input_grad[1]:
            dw = torch.zeros(CO, CI, ks, ks, dtype=torch.float32, device=x.device)
            ws = torch.empty(L.lib().p4c_conv_wgrad_workspace_bytes(64, ks) // 4, dtype=torch.float32, device=x.device)
            L.call("p4c_conv_wgrad_compact", L.ptr(x), C, ks, L.ptr(dy), CO, CO, CI, L.ptr(dw), L.ptr(ws), B, H, W, L.stream(x.device))
            dw = dw.to(w.dtype)
        return dx, dw


def _compact_ok(x: torch.Tensor, w: torch.Tensor) -> bool:
    CO, CI, ks, _ = w.shape
    B, H, W, C = x.shape
    return (x.dtype == torch.bfloat16 and C == CI and C % 8 == 0 and CO % 8 == 0 and C <= 64 and CO <= 64 and (C < 64 or CO < 64)
            and L.diag_switch("P4C_NO_COMPACT_CONV") != "1"
            and bool(L.lib().p4c_conv_compact_supported(C, CO, ks, B, H, W)))


def conv_nhwc_supported(x: torch.Tensor, w: torch.Tensor) -> bool:
    CO, CI, ks, ks2 = w.shape
    return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and CO <= 64 and ks == ks2 and ks in (1, 3) and CI <= 96
            and x.dim() == 4)


def conv_nhwc(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """x (B,H,W,CI) features-last, w (CO<=64,CI<=96,ks,ks) canonical torch weight -> (B,H,W,CO).  The kernels work on 32 / 64 / 96
    input channels and 64 output channels: the input is zero-padded to a multiple of 32 channels when needed (one copy), fewer
    output channels run as zero rows of a 64-channel launch and are sliced off (a view)."""
    CO, CI = w.shape[0], w.shape[1]
    if _compact_ok(x, w):
        return _ConvCompact.apply(x.contiguous(), w)
    cp = _pad32(CI)
    if x.shape[-1] != cp:
        x = torch.nn.functional.pad(x, (0, cp - x.shape[-1]))
    if CO < 64:
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, 64 - CO))
    compute = "bf16" if x.dtype == torch.bfloat16 else "f32"
    y = _ConvNHWC.apply(x.contiguous(), w, compute)
    return y if CO == 64 else y[..., :CO]


class _DeterministicSolvers:
    """``torch.backends.cudnn.deterministic = True`` for the duration of one library call.  The flag is process-global and the
    backward runs on autograd's worker thread: the toggle is serialised by a lock and counted, so that overlapping calls (a forward
    on the main thread, a backward of another graph on the engine thread) restore the HOST's value exactly once, after the last of
    them has left.  (A host convolution on a third thread may still observe the pinned flag while a call of ours is inside; set
    ``torch.backends.cudnn.deterministic`` yourself, as ``bench.py --deterministic`` does, to take the toggle out altogether.)"""

    _lock = threading.Lock()
    _depth = 0
    _keep = False

    def __enter__(self):
        cls = _DeterministicSolvers
        with cls._lock:
            if cls._depth == 0:
                cls._keep = torch.backends.cudnn.deterministic
                torch.backends.cudnn.deterministic = True
            cls._depth += 1

    def __exit__(self, *exc):
        cls = _DeterministicSolvers
        with cls._lock:
            cls._depth -= 1
            if cls._depth == 0:
                torch.backends.cudnn.deterministic = cls._keep
        return False


class _LibraryConv(torch.autograd.Function):
    """torch's convolution (MIOpen) for the shapes the MFMA kernels do not serve, pinned to the library's DETERMINISTIC solvers in the
    forward AND in both gradients.  The default solver choice accumulates with atomics whose order changes from run to run; with bf16
    activations that rounding noise moves LeakyReLU / softmax branches downstream, and two eager SwinUNetR steps on one batch and one
    set of weights differed by 19-26 % in their parameter gradients (UNetRPP: 20 %) while every native node was bit-identical.  With
    these few calls pinned the whole training step reproduces bit for bit (tools/diagnostics/determinism_probe.py,
    profiles/r03_determinism_probe.txt) at the same speed (SwinUNetR 38.6 vs 38.8 ms per step).  The flag is set around the call only
    -- the host application's own convolutions keep its setting."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, padding, dilation, groups):
        ctx.save_for_backward(x, w)
        ctx.conf = (tuple(stride), tuple(padding), tuple(dilation), int(groups), None if bias is None else list(bias.shape))
        with _DeterministicSolvers():
            return torch.nn.functional.conv2d(x, w, bias, stride, padding, dilation, groups)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, padding, dilation, groups, bias_sizes = ctx.conf
        mask = [ctx.needs_input_grad[0], ctx.needs_input_grad[1], bias_sizes is not None and ctx.needs_input_grad[2]]
        with _DeterministicSolvers():
            gx, gw, gb = torch.ops.aten.convolution_backward(gy, x, w, bias_sizes, list(stride), list(padding), list(dilation), False,
                                                             [0, 0], groups, mask)
        return gx, gw, gb, None, None, None, None


def library_conv2d(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, stride=(1, 1), padding=(0, 0), dilation=(1, 1),
                   groups: int = 1) -> torch.Tensor:
    """F.conv2d on the library's deterministic solvers (see _LibraryConv); string paddings ('same') are resolved by torch itself."""
    if isinstance(padding, str) or not x.is_cuda:
        return torch.nn.functional.conv2d(x, w, bias, stride, padding, dilation, groups)
    pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)
    return _LibraryConv.apply(x, w, bias, pair(stride), pair(padding), pair(dilation), groups)
