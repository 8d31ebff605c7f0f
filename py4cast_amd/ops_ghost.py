"""Autograd wrapper of the Ghost module's cheap operation (include/py4cast_hip.h: p4c_ghost_dw_*): depthwise 3x3 over the 32 primary
channels of a (B,H,W,64) features-last tensor, concatenated behind them.  No CPU fallback."""

import torch

from . import _lib as L


class _GhostDW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, w):
        L.require_cuda(y, w)
        B, H, W, C = y.shape
        if C != 64 or tuple(w.shape) != (32, 1, 3, 3):
            raise L.P4CError("ghost_dw: (B,H,W,64) tensor and a (32,1,3,3) depthwise weight")
        y = y.contiguous()
        wf = w.detach().float().reshape(32, 9).contiguous()
        out = torch.empty_like(y)
        L.call("p4c_ghost_dw_fwd", L.ptr(y), L.ptr(wf), L.ptr(out), L.dtype_code(y.dtype), B, H, W, L.stream(y.device))
        ctx.save_for_backward(y, wf)
        ctx.wdtype = w.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        y, wf = ctx.saved_tensors
        B, H, W, _ = y.shape
        dout = dout.contiguous()
        din = dw = None
        if ctx.needs_input_grad[0]:
            din = torch.empty_like(y)
            L.call("p4c_ghost_dw_bwd_data", L.ptr(dout), L.ptr(wf), L.ptr(din), L.dtype_code(y.dtype), B, H, W, L.stream(y.device))
        if ctx.needs_input_grad[1]:
            nb = L.lib().p4c_ghost_dw_wgrad_blocks(B, H, W)
            part = torch.empty(nb, 32, 9, dtype=torch.float32, device=y.device)
            L.call("p4c_ghost_dw_wgrad", L.ptr(y), L.ptr(dout), L.ptr(part), L.dtype_code(y.dtype), B, H, W, L.stream(y.device))
            dw = part.sum(dim=0).view(32, 1, 3, 3).to(ctx.wdtype)
        return din, dw


def ghost_dw(y: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """y (B,H,W,64) whose channels 0..31 hold the primary features (32..63 ignored) -> same tensor with channels 32..63 =
    depthwise3x3(y[..., :32]; w)."""
    return _GhostDW.apply(y, w)
