"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (SwinUNetR lives in mfai v5.0.1 / MONAI, absent here).

Torch-native restatement of the attention of one 2-D Swin block (Liu et al. 2021; the network behind
``model_name: SwinUNetR``, config/CLI/model/swinunetr.yaml:19-30), written op by op the way the published implementation
runs it: cyclic shift (torch.roll), window_partition, per-head q k^T * scale + relative position bias + shift mask (0 / -100),
softmax, @ v, window_reverse, reverse shift.
"""

import torch


def window_partition(x, ws):
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws * ws, C)


def window_reverse(windows, ws, B, H, W):
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def shift_mask(H, W, ws, shift, dtype=torch.float32):
    """(nW, N, N) additive mask of the shifted configuration (0 / -100), as Swin's compute_mask."""
    img = torch.zeros(1, H, W, 1, dtype=dtype)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).squeeze(-1)  # (nW, N)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def relative_position_index(ws):
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)  # (N, N)


def window_attention(qkv, bias, heads, ws, shift=0, scale=None):
    """qkv (B,H,W,3*C) -> (B,H,W,C); bias (heads,N,N) [query][key] or None."""
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    d = C // heads
    scale = d ** -0.5 if scale is None else scale
    x = torch.roll(qkv, shifts=(-shift, -shift), dims=(1, 2)) if shift > 0 else qkv
    xw = window_partition(x, ws)  # (B*nW, N, 3C)
    Bn, N, _ = xw.shape
    q, k, v = xw.reshape(Bn, N, 3, heads, d).permute(2, 0, 3, 1, 4)  # each (B*nW, heads, N, d)
    attn = (q * scale) @ k.transpose(-2, -1)
    if bias is not None:
        attn = attn + bias.unsqueeze(0)
    if shift > 0:
        m = shift_mask(H, W, ws, shift, attn.dtype).to(attn.device)  # (nW, N, N)
        nW = m.shape[0]
        attn = attn.view(Bn // nW, nW, heads, N, N) + m.unsqueeze(1).unsqueeze(0)
        attn = attn.view(Bn, heads, N, N)
    attn = attn.softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(Bn, N, C)
    out = window_reverse(out, ws, B, H, W)
    return torch.roll(out, shifts=(shift, shift), dims=(1, 2)) if shift > 0 else out
