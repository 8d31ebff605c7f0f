"""Autograd wrapper of the fused Swin window attention (include/py4cast_hip.h: p4c_window_attn_fwd/bwd).

Replaces, for one Swin block (SwinUNetR: config/CLI/model/swinunetr.yaml:19-30), the chain
roll -> window_partition -> qkv reshape/permute -> q@k^T*scale + relative_position_bias + attn_mask -> softmax -> @v ->
permute -> window_reverse -> roll back  by one kernel each way on the (B, Hp, Wp, 3*C) output of the qkv Linear.
No CPU fallback.
"""

from typing import Optional

import torch

from . import _lib as L


class _WindowAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, bias, heads: int, ws: int, shift: int, scale: float):
        L.require_cuda(qkv)
        qkv = qkv.contiguous()
        B, Hp, Wp, C3 = qkv.shape
        C = C3 // 3
        if C * 3 != C3 or C % heads:
            raise L.P4CError(f"window_attention: last dim {C3} is not 3 * heads * head_dim")
        d = C // heads
        bias_t = None if bias is None else bias.detach().to(torch.float32).transpose(1, 2).contiguous()  # [head][key][query]
        out = torch.empty(B, Hp, Wp, C, dtype=qkv.dtype, device=qkv.device)
        L.call("p4c_window_attn_fwd", L.ptr(qkv), L.ptr(bias_t), L.ptr(out), B, Hp, Wp, heads, d, ws, shift, float(scale),
               L.dtype_code(qkv.dtype), L.stream(qkv.device), alg_bytes=B * Hp * Wp * 4 * C * qkv.element_size())
        ctx.save_for_backward(qkv, bias_t)
        ctx.cfg = (heads, d, ws, shift, float(scale), None if bias is None else bias.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, bias_t = ctx.saved_tensors
        heads, d, ws, shift, scale, bias_dtype = ctx.cfg
        B, Hp, Wp, _ = qkv.shape
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        want_dbias = bias_t is not None and ctx.needs_input_grad[1]
        dbias_t = torch.empty_like(bias_t) if want_dbias else None
        wsp = None
        if want_dbias:
            nbytes = L.lib().p4c_window_attn_bwd_workspace_bytes(B, Hp, Wp, heads, ws)
            wsp = torch.empty(nbytes // 4, dtype=torch.float32, device=qkv.device)
        L.call("p4c_window_attn_bwd", L.ptr(qkv), L.ptr(bias_t), L.ptr(dout), L.ptr(dqkv), L.ptr(dbias_t), L.ptr(wsp), B, Hp, Wp,
               heads, d, ws, shift, scale, L.dtype_code(qkv.dtype), L.stream(qkv.device),
               alg_bytes=B * Hp * Wp * 7 * heads * d * qkv.element_size())
        dbias = dbias_t.transpose(1, 2).to(bias_dtype) if want_dbias else None
        return dqkv, dbias, None, None, None, None


def window_attention(qkv: torch.Tensor, bias: Optional[torch.Tensor], heads: int, ws: int, shift: int = 0,
                     scale: Optional[float] = None) -> torch.Tensor:
    """qkv (B,Hp,Wp,3*heads*head_dim) [q|k|v, each (heads, head_dim)], bias (heads, ws*ws, ws*ws) [query][key] or None
    -> (B,Hp,Wp,heads*head_dim).  Hp, Wp multiples of ws (pad first, as Swin does); head_dim in {8,16,32}."""
    d = qkv.shape[-1] // 3 // heads
    return _WindowAttention.apply(qkv, bias, heads, ws, shift, d ** -0.5 if scale is None else scale)
