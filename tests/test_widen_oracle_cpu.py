"""CPU checks of the oracles of the widened model kernels (mesh-GNN edge passes, Swin window attention): each restatement is
compared with an independent formulation of the same published operation (parity with mfai itself is unpinned: the package is
absent from the reference checkout, see oracle/__init__.py)."""

import math

import pytest
import torch

from oracle import graph as og
from oracle import window_attention as owa


def test_edge_first_layer_distributes_over_concat():
    torch.manual_seed(0)
    E, Ns, Nr, C = 200, 30, 17, 16
    e, xs, xr = torch.randn(E, C, dtype=torch.float64), torch.randn(Ns, C, dtype=torch.float64), torch.randn(Nr, C, dtype=torch.float64)
    src, dst = torch.randint(0, Ns, (E,)), torch.randint(0, Nr, (E,))
    w, bias = torch.randn(C, 3 * C, dtype=torch.float64), torch.randn(C, dtype=torch.float64)
    ref = og.edge_mlp_first_layer_concat(e, xs, xr, src, dst, w, bias, "silu")
    got = og.edge_gather_add(e @ w[:, :C].T + bias, xs @ w[:, C:2 * C].T, src, xr @ w[:, 2 * C:].T, dst, "silu")
    assert torch.allclose(got, ref, rtol=1e-12, atol=1e-12)


def test_aggregate_sum_matches_loop():
    torch.manual_seed(1)
    E, N, C = 64, 9, 4
    msg, dst = torch.randn(E, C, dtype=torch.float64), torch.randint(0, N, (E,))
    ref = torch.zeros(N, C, dtype=torch.float64)
    for i in range(E):
        ref[dst[i]] += msg[i]
    assert torch.allclose(og.aggregate_sum(msg, dst, N), ref, rtol=1e-12, atol=1e-12)


def _attention_per_pixel(qkv, bias, heads, ws, shift, scale):
    """Brute force: for every pixel find its window in the rolled grid, attend over that window token by token."""
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    d = C // heads
    out = torch.zeros(B, H, W, C, dtype=qkv.dtype)

    def region(v, size):  # wrap-around region of a ROLLED coordinate
        return 0 if v < size - ws else (1 if v < size - shift else 2)

    for b in range(B):
        for yr in range(H):          # rolled coordinates
            for xr in range(W):
                y, x = (yr + shift) % H, (xr + shift) % W
                wy0, wx0 = yr // ws * ws, xr // ws * ws
                qi = (yr - wy0) * ws + (xr - wx0)
                for hd in range(heads):
                    q = qkv[b, y, x, hd * d:(hd + 1) * d]
                    logits, vals = [], []
                    for ky in range(ws):
                        for kx in range(ws):
                            kyr, kxr = wy0 + ky, wx0 + kx
                            yy, xx = (kyr + shift) % H, (kxr + shift) % W
                            k = qkv[b, yy, xx, C + hd * d:C + (hd + 1) * d]
                            s = float(q @ k) * scale
                            if bias is not None:
                                s += float(bias[hd, qi, ky * ws + kx])
                            if shift > 0 and (region(yr, H), region(xr, W)) != (region(kyr, H), region(kxr, W)):
                                s += -100.0
                            logits.append(s)
                            vals.append(qkv[b, yy, xx, 2 * C + hd * d:2 * C + (hd + 1) * d])
                    p = torch.softmax(torch.tensor(logits, dtype=qkv.dtype), 0)
                    out[b, y, x, hd * d:(hd + 1) * d] = (p[:, None] * torch.stack(vals)).sum(0)
    return out


@pytest.mark.parametrize("ws,shift", [(4, 0), (4, 2), (3, 1)])
def test_window_attention_oracle_matches_per_pixel(ws, shift):
    torch.manual_seed(2)
    B, H, W, heads, d = 1, 2 * ws, 3 * ws, 2, 4
    qkv = torch.randn(B, H, W, 3 * heads * d, dtype=torch.float64)
    bias = torch.randn(heads, ws * ws, ws * ws, dtype=torch.float64)
    ref = _attention_per_pixel(qkv, bias, heads, ws, shift, d ** -0.5)
    got = owa.window_attention(qkv, bias, heads, ws, shift)
    assert torch.allclose(got, ref, rtol=1e-10, atol=1e-10)


def test_relative_position_index_is_toeplitz():
    ws = 4
    idx = owa.relative_position_index(ws)
    assert idx.shape == (16, 16) and idx.min() == 0 and idx.max() == (2 * ws - 1) ** 2 - 1
    for a in range(16):
        for b in range(16):
            dy, dx = a // ws - b // ws, a % ws - b % ws
            assert idx[a, b] == (dy + ws - 1) * (2 * ws - 1) + dx + ws - 1


def test_unetrpp_oracle_epa_matches_brute_force_per_head():
    """oracle/unetrpp.py::EPA (batched transposes / matmuls) against a per-(sample, head) loop that builds every matrix from its
    definition: column-normalised q and k, d x d channel attention, token-axis projection E = F, N x p spatial attention."""
    from oracle.unetrpp import EPA, UNetRPP

    torch.manual_seed(0)
    B, N, C, h, p = 2, 48, 16, 4, 8
    d = C // h
    e = EPA(N, C, p, h).double()
    with torch.no_grad():
        e.temperature.uniform_(0.5, 1.5)
        e.temperature2.uniform_(0.5, 1.5)
        x = torch.randn(B, N, C, dtype=torch.float64)
        y = e(x)
        qkvv = (x @ e.qkvv.weight.t()).view(B, N, 4, h, d)
        sa, ca = torch.zeros(B, N, C, dtype=torch.float64), torch.zeros(B, N, C, dtype=torch.float64)
        for b in range(B):
            for hh in range(h):
                q, k, v1, v2 = (qkvv[b, :, i, hh] for i in range(4))
                qn, kn = q / q.norm(dim=0).clamp_min(1e-12), k / k.norm(dim=0).clamp_min(1e-12)
                A = torch.softmax(qn.t() @ kn * e.temperature[hh, 0, 0], -1)
                ca[b, :, hh * d:(hh + 1) * d] = v1 @ A.t()
                KP = (e.E.weight @ k + e.E.bias[:, None]).t()
                VP = (e.E.weight @ v2 + e.E.bias[:, None]).t()
                S = torch.softmax(qn @ KP * e.temperature2[hh, 0, 0], -1)
                sa[b, :, hh * d:(hh + 1) * d] = S @ VP.t()
        ref = torch.cat([sa @ e.out_proj.weight.t() + e.out_proj.bias, ca @ e.out_proj2.weight.t() + e.out_proj2.bias], -1)
    assert float((y - ref).abs().max()) < 1e-12
    net = UNetRPP(7, 3, (64, 64), hidden_size=64, num_heads_encoder=4, num_heads_decoder=4)
    out = net(torch.randn(1, 64, 64, 7))
    assert out.shape == (1, 64, 64, 3) and bool(torch.isfinite(out).all())


def test_unetrpp_oracle_published_block():
    """oracle/unetrpp.py with published_block=True: the EPA against a per-(sample, head) loop whose spatial branch is merged exactly
    as the published code writes it -- ``(attn_SA @ v_SA^T).permute(0, 3, 1, 2).reshape(B, N, C)`` --, the published state-dict keys
    (conv8.1.*, E.* and F.* for ONE Linear), the conv8 channel dropout drawn in training mode only."""
    from oracle.unetrpp import EPA, TransformerBlock, UNetRPP

    torch.manual_seed(1)
    B, N, C, h, p = 2, 48, 16, 4, 8
    d = C // h
    e = EPA(N, C, p, h, published=True).double()
    assert e.F is e.E and {"E.weight", "E.bias", "F.weight", "F.bias"} <= set(e.state_dict())
    with torch.no_grad():
        e.temperature.uniform_(0.5, 1.5)
        e.temperature2.uniform_(0.5, 1.5)
        x = torch.randn(B, N, C, dtype=torch.float64)
        y = e(x)
        qkvv = (x @ e.qkvv.weight.t()).view(B, N, 4, h, d)
        T = torch.zeros(B, h, N, d, dtype=torch.float64)
        ca = torch.zeros(B, N, C, dtype=torch.float64)
        for b in range(B):
            for hh in range(h):
                q, k, v1, v2 = (qkvv[b, :, i, hh] for i in range(4))
                qn, kn = q / q.norm(dim=0).clamp_min(1e-12), k / k.norm(dim=0).clamp_min(1e-12)
                ca[b, :, hh * d:(hh + 1) * d] = v1 @ torch.softmax(qn.t() @ kn * e.temperature[hh, 0, 0], -1).t()
                KP = (e.E.weight @ k + e.E.bias[:, None]).t()
                VP = (e.F.weight @ v2 + e.F.bias[:, None]).t()
                T[b, hh] = torch.softmax(qn @ KP * e.temperature2[hh, 0, 0], -1) @ VP.t()
        sa = T.permute(0, 3, 1, 2).reshape(B, N, C)          # the published merge
        ref = torch.cat([sa @ e.out_proj.weight.t() + e.out_proj.bias, ca @ e.out_proj2.weight.t() + e.out_proj2.bias], -1)
        assert float((y - ref).abs().max()) < 1e-12
        assert float((y - EPA(N, C, p, h).double()(x)).abs().max()) > 0      # (sanity: the restated module is another object / function)
    blk = TransformerBlock(64, 16, 8, 4, published=True, conv8_dropout=0.1)
    assert isinstance(blk.conv8, torch.nn.Sequential) and isinstance(blk.conv8[0], torch.nn.Dropout2d) and blk.conv8[0].p == 0.1
    assert {"conv8.1.weight", "conv8.1.bias"} <= set(blk.state_dict()) and "conv8.weight" not in blk.state_dict()
    net = UNetRPP(7, 3, (64, 64), hidden_size=64, num_heads_encoder=4, num_heads_decoder=4)      # defaults: published, p = 0.1
    xin = torch.randn(1, 64, 64, 7)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if n_.endswith("gamma"):
                p_.fill_(0.5)
        net.eval()
        a, b = net(xin), net(xin)
        net.train()
        torch.manual_seed(3); c = net(xin)
        torch.manual_seed(4); dd = net(xin)
    assert torch.equal(a, b) and not torch.equal(c, dd)
    old = UNetRPP(7, 3, (64, 64), hidden_size=64, num_heads_encoder=4, num_heads_decoder=4, published_block=False)
    assert "stages.0.0.conv8.weight" in old.state_dict() and not any(".F." in k for k in old.state_dict())


def test_static_index_gather_backward_matches_index_put():
    """swinunetr._TableRows (the relative-position-bias gather with a fixed-order backward over the inverse of its static index)
    against the plain ``table[index]`` and autograd's index_put backward, for Swin's window sizes."""
    import torch

    from py4cast_amd.swinunetr import _TableRows, inverse_index_table, relative_position_index

    for ws, heads in ((7, 3), (4, 6), (2, 1)):
        idx = relative_position_index(ws).view(-1)
        rows = (2 * ws - 1) ** 2
        inv = inverse_index_table(idx, rows)
        assert inv.shape[0] == rows and int((inv < idx.numel()).sum()) == idx.numel()          # every gathered row listed exactly once
        table = torch.randn(rows, heads, dtype=torch.float64, requires_grad=True)
        ref = table.detach().clone().requires_grad_(True)
        g = torch.randn(idx.numel(), heads, dtype=torch.float64)
        y = _TableRows.apply(table, idx, inv)
        y.backward(g)
        yr = ref[idx]
        yr.backward(g)
        assert torch.equal(y, yr)
        assert float((table.grad - ref.grad).abs().max()) < 1e-12
