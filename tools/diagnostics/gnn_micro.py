"""Micro-benchmark of the mesh-GNN / Swin kernels at the sizes of the 512x512 benchmark configuration (B = 2):
rows of 64 bf16 features; E = 2.1 M (mesh->grid edges), N = 0.52 M grid nodes.  Prints the HIP-event time and the achieved
algorithmic GB/s per kernel; run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (one counter per pass) with `--once`
to get the HBM traffic per launch (profiles/r01_pmc_gnn.json)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py4cast_amd import _lib as L  # noqa: E402
from py4cast_amd import ops_graph as G  # noqa: E402
from py4cast_amd import ops_rows as R  # noqa: E402
from py4cast_amd.ops_attention import window_attention  # noqa: E402
from py4cast_amd.ops_mlp import row_mlp  # noqa: E402

once = "--once" in sys.argv
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, n_grid, n_mesh, C = 2, 512 * 512, 6561, 64
E = 4 * n_grid
dst = torch.arange(n_grid).repeat_interleave(4)
src = torch.randint(0, n_mesh, (E,))
es = G.EdgeSet(torch.cat([src, src + n_mesh]), torch.cat([dst, dst + n_grid]), B * n_mesh, B * n_grid).to(dev)
E2 = B * E
bf = torch.bfloat16
base = torch.randn(E2, C, device=dev).to(bf)
base2 = torch.randn(E2, C, device=dev).to(bf)
a = torch.randn(B * n_mesh, C, device=dev).to(bf)
b = torch.randn(B * n_grid, C, device=dev).to(bf)
gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
w = torch.randn(64, 64, device=dev) * 0.1
qkv = torch.randn(2, 259, 259, 72, device=dev).to(bf)
bias = torch.randn(3, 49, 49, device=dev)
dout = torch.randn(2, 259, 259, 24, device=dev).to(bf)


def wgrad():
    buf = torch.empty(64 * 64 + 64, dtype=torch.float32, device=dev)
    ws = torch.empty(L.lib().p4c_row_linear_wgrad_workspace_bytes(E2, 64) // 4, dtype=torch.float32, device=dev)
    return lambda: L.call("p4c_row_linear_wgrad", L.ptr(base), L.ptr(base2), L.ptr(buf), L.ptr(ws), E2, 64, 64, L.BF16, L.stream(dev))


def ln_bwd():
    dx = torch.empty_like(base)
    dgb = torch.empty(2, C, dtype=torch.float32, device=dev)
    ws = torch.empty(L.lib().p4c_row_layernorm_bwd_workspace_bytes(E2, C, L.BF16) // 4, dtype=torch.float32, device=dev)
    return lambda: L.call("p4c_row_layernorm_bwd", L.ptr(base), L.ptr(base2), L.ptr(gamma), 1e-5, L.ptr(dx), L.ptr(dgb), L.ptr(dgb[1]),
                          L.ptr(ws), E2, C, L.BF16, L.stream(dev))


def attn_bwd():
    q = qkv.clone().requires_grad_(True)
    bb = bias.clone().requires_grad_(True)
    out = window_attention(q, bb, 3, 7, 3)
    return lambda: torch.autograd.grad(out, (q, bb), dout, retain_graph=True)


def mlp_pair():
    """fused edge MLP of the mesh->grid InteractionNet: forward, and forward + backward (the difference is the backward)"""
    w1 = (torch.randn(64, 192, device=dev) * 0.1).requires_grad_(True)
    b1, b2 = [(torch.randn(64, device=dev) * 0.1).requires_grad_(True) for _ in range(2)]
    w2 = (torch.randn(64, 64, device=dev) * 0.1).requires_grad_(True)
    g, bt = (torch.rand(64, device=dev) + 0.5).requires_grad_(True), torch.zeros(64, device=dev, requires_grad=True)
    e = base.clone().requires_grad_(True)
    aa, bb = a.clone().requires_grad_(True), b.clone().requires_grad_(True)

    def fwd():
        return row_mlp(e, w1[:, :64], b1, w2, b2, g, bt, 1e-5, ga=aa, gb=bb, edges=es, res=e)

    def fwd_bwd():
        msg, new_e = fwd()
        torch.autograd.grad((msg, new_e), (e, aa, bb, w1, w2), (base2, base2))

    return fwd, fwd_bwd


mlp_fwd, mlp_fwd_bwd = mlp_pair()
row = C * 2
cases = {
    "row_mlp_fwd (m2g edge MLP: gathers, LN, msg + residual out)": (mlp_fwd, E2 * row * 4 + (B * n_mesh + B * n_grid) * row + 8 * E2),
    "row_mlp fwd + bwd (+ 2 segment sums of dpre)": (mlp_fwd_bwd, E2 * row * 4 + E2 * row * 5 + 2 * (B * n_mesh + B * n_grid) * row + 16 * E2
                                                     + 2 * E2 * row),
    "edge_gather_add_fwd (m2g, silu)": (lambda: G._gather_raw(base, a, es.src, b, es.dst, None, 2, E2, C, base),
                                        2 * E2 * row + (B * n_mesh + B * n_grid) * row + 8 * E2),
    "edge_gather_add_bwd (m2g, silu)": (lambda: G._gather_raw(base, a, es.src, b, es.dst, base2, 2, E2, C, base),
                                        3 * E2 * row + (B * n_mesh + B * n_grid) * row + 8 * E2),
    "segment_sum (m2g: 4 edges per grid node)": (lambda: G._segment_sum_raw(base, *es.by_dst, B * n_grid),
                                                 E2 * row + 4 * E2 + B * n_grid * (row + 4)),
    "segment_sum (adjoint by sender: 320 edges per mesh node)": (lambda: G._segment_sum_raw(base, *es.by_src, B * n_mesh),
                                                                  E2 * row + 4 * E2 + B * n_mesh * (row + 4)),
    "row_layernorm_fwd (+res)": (lambda: R.row_layer_norm(base, gamma, beta, 1e-5, base2), 3 * E2 * row),
    "row_layernorm_bwd": (ln_bwd(), 3 * E2 * row),
    "row_linear_wgrad (K=64)": (wgrad(), 2 * E2 * row),
    "window_attn_fwd (259x259 tokens, 3 heads x 8, ws 7, shift 3)": (lambda: window_attention(qkv, bias, 3, 7, 3),
                                                                      2 * 259 * 259 * 96 * 2),
    "window_attn_bwd": (attn_bwd(), 2 * 259 * 259 * (72 + 24 + 72) * 2),
}
for name, (fn, nbytes) in cases.items():
    fn()
    torch.cuda.synchronize()
    if once:
        fn()
        torch.cuda.synchronize()
        continue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:62s} {us:8.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / us / 1e3:7.0f} GB/s")
