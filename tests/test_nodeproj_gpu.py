"""GPU parity of the mesh-GNN launch grouping (csrc/nodeproj.hip, py4cast_amd/ops_nodeproj.py): the node projections of an
InteractionNet (config/CLI/model/graphlam.yaml:19-26; the distributed first Linears of py4cast_amd/graphlam.py) as one launch per
direction, against float64 on the same bf16 operands, and the batched reduction queue against reducing after every call (bit for bit)."""

import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _weights(device, seed):
    torch.manual_seed(seed)
    wide = (torch.randn(64, 192, device=device) * 0.1).requires_grad_(True)     # an edge MLP's first Linear: [e | x_s | x_r]
    aggr = (torch.randn(64, 128, device=device) * 0.1).requires_grad_(True)     # a node-update MLP's first Linear: [x_r | agg]
    return wide, aggr


@pytest.mark.parametrize("n", [1, 2, 3])
@pytest.mark.parametrize("R", [1, 18, 162, 1458, 2049, 13122, 70001])
def test_node_proj_against_float64(gpu_device, R, n):
    """outputs (one bf16 rounding), input gradient (one rounding of the K = 64 n sum) and the weight gradients added into the .grad views
    (fp32 accumulation: <= 5e-4) against float64 on the same bf16 operands; the other columns of the wide gradients stay untouched"""
    from py4cast_amd.ops_nodeproj import node_proj

    wide, aggr = _weights(gpu_device, 11)
    wide.grad, aggr.grad = torch.full_like(wide, 0.25), torch.full_like(aggr, -0.5)
    torch.manual_seed(12 + R)
    x = torch.randn(R, 64, device=gpu_device).bfloat16().requires_grad_(True)
    blocks = [wide[:, 64:128], wide[:, 128:], aggr[:, :64]][:n]
    cots = [torch.randn(R, 64, device=gpu_device).bfloat16() for _ in range(n)]
    ys = node_proj(x, blocks)
    assert len(ys) == n and all(y.dtype == torch.bfloat16 and y.shape == (R, 64) for y in ys)
    sum((y.float() * c.float()).sum() for y, c in zip(ys, cots)).backward()
    xd = x.detach().double()
    for y, w in zip(ys, blocks):
        assert _rel(y, xd @ w.detach().bfloat16().double().t()) < 4e-3
    dx_ref = sum(c.double() @ w.detach().bfloat16().double() for c, w in zip(cots, blocks))
    assert _rel(x.grad, dx_ref) < 4e-3
    gw = [wide.grad[:, 64:128] - 0.25, wide.grad[:, 128:] - 0.25, aggr.grad[:, :64] + 0.5]
    for i in range(n):
        ref = cots[i].double().t() @ xd
        assert _rel(gw[i], ref) < 5e-4, (i, _rel(gw[i], ref))
    assert float((wide.grad[:, :64] - 0.25).abs().max()) == 0.0 and float((aggr.grad[:, 64:] + 0.5).abs().max()) == 0.0
    for i in range(n, 3):       # projections that were not asked for: untouched
        assert float(gw[i].abs().max()) == 0.0


def test_node_proj_equals_separate_row_linears(gpu_device):
    """against the route it replaces (ops_rows.row_linear per block): identical outputs (the same MFMA chain per element), input gradient
    within a bf16 rounding of the sum, weight gradients within fp32 summation order"""
    from py4cast_amd.ops_nodeproj import node_proj
    from py4cast_amd.ops_rows import row_linear

    torch.manual_seed(21)
    x = torch.randn(5000, 64, device=gpu_device).bfloat16().requires_grad_(True)
    cot = [torch.randn(5000, 64, device=gpu_device).bfloat16() for _ in range(3)]

    def run(grouped):
        wide, aggr = _weights(gpu_device, 22)
        wide.grad, aggr.grad, x.grad = torch.zeros_like(wide), torch.zeros_like(aggr), None
        ws = [wide[:, 64:128], wide[:, 128:], aggr[:, :64]]
        ys = node_proj(x, ws) if grouped else [row_linear(x, w, grads_in_place=True) for w in ws]
        sum((y.float() * c.float()).sum() for y, c in zip(ys, cot)).backward()
        return [y.detach().clone() for y in ys], x.grad.clone(), wide.grad.clone(), aggr.grad.clone()

    ys_g, dx_g, gw_g, ga_g = run(True)
    ys_s, dx_s, gw_s, ga_s = run(False)
    for a, b in zip(ys_g, ys_s):
        assert torch.equal(a, b)
    assert _rel(dx_g, dx_s) < 1e-2
    assert _rel(gw_g, gw_s) < 1e-5 and _rel(ga_g, ga_s) < 1e-5


def test_node_proj_without_gradient_buffers_takes_the_autograd_route(gpu_device):
    from py4cast_amd.ops_nodeproj import node_proj

    wide, aggr = _weights(gpu_device, 31)
    torch.manual_seed(32)
    x = torch.randn(700, 64, device=gpu_device).bfloat16().requires_grad_(True)
    ys = node_proj(x, [wide[:, 64:128], aggr[:, :64]])
    (ys[0].float().sum() + 2 * ys[1].float().sum()).backward()
    xd = x.detach().double()
    assert _rel(wide.grad[:, 64:128], torch.ones(700, 64, dtype=torch.float64, device=gpu_device).t() @ xd) < 1e-2
    assert float(wide.grad[:, :64].abs().max()) == 0.0
    with pytest.raises(Exception):
        node_proj(x.cpu(), [wide[:, 64:128]])


def _stack(gpu_device, deferred, seed=41, reuse=False):
    """three 'AR steps' of projection -> fused MLP -> projection on shared parameters; returns every accumulated gradient"""
    from py4cast_amd import ops_nodeproj as NP
    from py4cast_amd.ops_mlp import row_mlp

    wide, aggr = _weights(gpu_device, seed)
    torch.manual_seed(seed + 1)
    b1, b2, beta = [(torch.randn(64, device=gpu_device) * 0.1).requires_grad_(True) for _ in range(3)]
    w2 = (torch.randn(64, 64, device=gpu_device) * 0.1).requires_grad_(True)
    gamma = (torch.rand(64, device=gpu_device) + 0.5).requires_grad_(True)
    params = [wide, aggr, b1, w2, b2, gamma, beta]
    for p in params:
        p.grad = torch.zeros_like(p)
    x = torch.randn(1458, 64, device=gpu_device).bfloat16().requires_grad_(True)
    NP.GradQueue.enabled = deferred
    try:
        loss = 0.0
        h = x
        for _ in range(3):
            a, part = NP.node_proj(h, [wide[:, 64:128], aggr[:, :64]])
            _, h = row_mlp(a, wide[:, :64], b1, w2, b2, gamma, beta, 1e-5, ga=part, res=h, want_out=False, grads_in_place=True)
            if reuse:   # the same block twice inside one pass: the queue must keep the two additions in order
                h2, = NP.node_proj(h, [wide[:, 64:128]])
                h = h + h2
            loss = loss + h.float().square().mean()
        loss.backward()
    finally:
        NP.GradQueue.enabled = True
    return [p.grad.clone() for p in params] + [x.grad.clone()]


@pytest.mark.parametrize("reuse", [False, True])
def test_deferred_reduction_is_bit_identical(gpu_device, reuse):
    from py4cast_amd import _lib as L

    ref = _stack(gpu_device, False, reuse=reuse)
    got = _stack(gpu_device, True, reuse=reuse)
    again = _stack(gpu_device, True, reuse=reuse)
    for a, b, c in zip(got, ref, again):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert float(ref[0].abs().sum()) > 0
    assert L.lib().p4c_grad_reduce_pending() == 0


def test_queue_flushes_more_jobs_than_one_launch_holds(gpu_device):
    """40 independent projections in one backward pass (> 32 jobs per launch) + a flush through the C entry points directly"""
    from py4cast_amd import _lib as L
    from py4cast_amd.ops_nodeproj import node_proj

    torch.manual_seed(51)
    x = torch.randn(300, 64, device=gpu_device).bfloat16()
    ws = [(torch.randn(64, 64, device=gpu_device) * 0.1).requires_grad_(True) for _ in range(40)]
    for w in ws:
        w.grad = torch.zeros_like(w)
    sum(node_proj(x, [w])[0].float().sum() for w in ws).backward()
    ref = torch.ones(300, 64, dtype=torch.float64, device=gpu_device).t() @ x.double()
    for w in ws:
        assert _rel(w.grad, ref) < 5e-4
    lib = L.lib()
    assert lib.p4c_grad_reduce_pending() == 0
    assert lib.p4c_grad_reduce_defer(1) == 0 and lib.p4c_grad_reduce_defer(-1) == 1 and lib.p4c_grad_reduce_defer(0) == 0
    L.check(lib.p4c_grad_reduce_flush(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))


def test_node_proj_passthrough_sums_the_residual_gradient_inside_the_launch(gpu_device):
    """passthrough=True hands x back as one more output; a gradient arriving there (the node update's residual) is added inside the
    data-gradient launch: x.grad = sum_i dy_i W_i + d(res) against float64, and equals the route without passthrough up to the bf16
    rounding of the intermediate sum"""
    from py4cast_amd.ops_nodeproj import node_proj

    wide, aggr = _weights(gpu_device, 61)
    wide.grad, aggr.grad = torch.zeros_like(wide), torch.zeros_like(aggr)
    torch.manual_seed(62)
    R = 1458
    x = torch.randn(R, 64, device=gpu_device).bfloat16().requires_grad_(True)
    blocks = [wide[:, 128:], aggr[:, :64]]
    cots = [torch.randn(R, 64, device=gpu_device).bfloat16() for _ in range(3)]
    b, part, xr = node_proj(x, blocks, passthrough=True)
    assert xr.data_ptr() == x.data_ptr() and xr.requires_grad
    ((b.float() * cots[0].float()).sum() + (part.float() * cots[1].float()).sum() + (xr.float() * cots[2].float()).sum()).backward()
    ref = cots[0].double() @ blocks[0].detach().bfloat16().double() + cots[1].double() @ blocks[1].detach().bfloat16().double() + cots[2].double()
    assert _rel(x.grad, ref) < 4e-3
    got = x.grad.clone()
    x.grad = None
    b, part = node_proj(x, blocks)
    ((b.float() * cots[0].float()).sum() + (part.float() * cots[1].float()).sum() + (x.float() * cots[2].float()).sum()).backward()
    assert _rel(got, x.grad) < 1e-2
    # only the residual gradient arrives: it passes through unchanged
    x.grad = None
    _, _, xr = node_proj(x, blocks, passthrough=True)
    (xr.float() * cots[2].float()).sum().backward()
    assert torch.equal(x.grad, cots[2])


@pytest.mark.parametrize("gather", [False, True])
def test_row_mlp_with_res_is_x_folds_both_gradients_into_dx(gpu_device, gather):
    """an edge update e <- e + MLP(e, ...): res IS x, the kernel stores dx + dy_res (p4c_row_mlp_desc.dx_plus_dy_res); equal to the two
    separate gradients summed by autograd up to one bf16 rounding, parameter gradients bit-identical"""
    from py4cast_amd import ops_graph as G
    from py4cast_amd.ops_mlp import row_mlp

    torch.manual_seed(71)
    R, N = 3001, 257
    e0 = torch.randn(R, 64, device=gpu_device).bfloat16()
    ga = torch.randn(N, 64, device=gpu_device).bfloat16()
    src, dst = torch.randint(0, N, (R,)), torch.randint(0, N, (R,))
    es = G.EdgeSet(src, dst, N, N).to(gpu_device) if gather else None
    cot = [torch.randn(R, 64, device=gpu_device).bfloat16() for _ in range(2)]

    def run(same):
        wide, _ = _weights(gpu_device, 72)
        torch.manual_seed(73)
        b1, b2, beta = [(torch.randn(64, device=gpu_device) * 0.1).requires_grad_(True) for _ in range(3)]
        w2 = (torch.randn(64, 64, device=gpu_device) * 0.1).requires_grad_(True)
        gamma = (torch.rand(64, device=gpu_device) + 0.5).requires_grad_(True)
        params = [wide, b1, w2, b2, gamma, beta]
        for p in params:
            p.grad = torch.zeros_like(p)
        e = e0.clone().requires_grad_(True)
        r = e if same else e0.clone().requires_grad_(True)
        kw = dict(ga=ga, gb=ga, edges=es) if gather else {}
        msg, new_e = row_mlp(e, wide[:, :64], b1, w2, b2, gamma, beta, 1e-5, res=r, grads_in_place=True, **kw)
        ((msg.float() * cot[0].float()).sum() + (new_e.float() * cot[1].float()).sum()).backward()
        total = e.grad.float() if same else e.grad.float() + r.grad.float()
        return msg.detach(), new_e.detach(), total, [p.grad.clone() for p in params]

    m1, n1, g1, p1 = run(True)
    m0, n0, g0, p0 = run(False)
    assert torch.equal(m1, m0) and torch.equal(n1, n0)
    assert _rel(g1, g0) < 4e-3
    for a, b in zip(p1, p0):
        assert torch.equal(a, b)
