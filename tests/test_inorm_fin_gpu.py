"""The instance-norm statistics finished inside the reduce launch (csrc/inorm.hip: InFin, round 6; SwinUNETR's decoder blocks,
config/CLI/model/swinunetr.yaml:23 `norm_name: instance`, UNETR++'s full-resolution blocks and batch norms) -- measured SLOWER than the
reduce + finalize launches in the step and therefore not the product route (ops_inorm.FUSED_FINALIZE = False), but the entry points are
part of the C ABI: bit-identical to the two-launch route, against torch's InstanceNorm2d in float64, many launches back to back (ticket
reuse), and inside a HIP-graph capture."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(x, w, b, res, gy, fused):
    from py4cast_amd import ops_inorm as ON

    ON.FUSED_FINALIZE = fused
    try:
        xg = x.clone().requires_grad_(True)
        wg, bg = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        rg = None if res is None else res.clone().requires_grad_(True)
        y = ON.instance_norm_act(xg, wg, bg, 1e-5, 0.01, rg)
        y.backward(gy)
        return [y.detach(), xg.grad, wg.grad, bg.grad] + ([] if res is None else [rg.grad])
    finally:
        ON.FUSED_FINALIZE = False


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C,with_res", [(2, 64, 64, 24, True), (1, 17, 9, 48, False), (3, 8, 8, 384, True), (2, 128, 128, 64, False), (2, 5, 7, 1024, True)])
def test_fused_finalize_equals_the_two_launch_route_and_torch(gpu_device, dtype, B, H, W, C, with_res):
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, H, W, C, generator=g).to(gpu_device).to(dtype)
    w = (torch.rand(C, generator=g) + 0.5).to(gpu_device)
    b = torch.randn(C, generator=g).to(gpu_device) * 0.2
    res = torch.randn(B, H, W, C, generator=g).to(gpu_device).to(dtype) if with_res else None
    gy = torch.randn(B, H, W, C, generator=g).to(gpu_device).to(dtype)
    fused = _run(x, w, b, res, gy, True)
    plain = _run(x, w, b, res, gy, False)
    for a, c in zip(fused, plain):
        assert torch.equal(a, c)                      # the same sums in the same order
    for _ in range(40):                               # tickets are reset by the launch that used them: many launches in a row
        again = _run(x, w, b, res, gy, True)
    for a, c in zip(fused, again):
        assert torch.equal(a, c)
    # float64 reference
    xd = x.double().cpu().requires_grad_(True)
    wd, bd = w.double().cpu().requires_grad_(True), b.double().cpu().requires_grad_(True)
    z = torch.nn.functional.instance_norm(xd.permute(0, 3, 1, 2), weight=wd, bias=bd, eps=1e-5).permute(0, 2, 3, 1)
    if res is not None:
        z = z + res.double().cpu()
    yr = torch.nn.functional.leaky_relu(z, 0.01)
    yr.backward(gy.double().cpu())
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    rel = lambda a, r: float((a.double().cpu() - r).norm() / r.norm().clamp_min(1e-30))   # noqa: E731
    assert rel(fused[0], yr.detach()) < tol and rel(fused[1], xd.grad) < 4 * tol
    assert rel(fused[2], wd.grad) < 4 * tol and rel(fused[3], bd.grad) < 4 * tol


def test_fused_finalize_inside_a_hip_graph(gpu_device):
    """captured launches keep their ticket addresses; replays give the eager result bit for bit"""
    from py4cast_amd import ops_inorm as ON

    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 48, 24, generator=g).to(gpu_device).bfloat16()
    w = (torch.rand(24, generator=g) + 0.5).to(gpu_device)
    b = torch.randn(24, generator=g).to(gpu_device)
    gy = torch.randn(2, 32, 48, 24, generator=g).to(gpu_device).bfloat16()
    eager = _run(x, w, b, None, gy, True)
    xs = x.clone().requires_grad_(True)
    ws, bs = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ON.FUSED_FINALIZE = True
        try:
            for _ in range(2):
                ON.instance_norm_act(xs, ws, bs, 1e-5, 0.01, None).backward(gy)
        finally:
            ON.FUSED_FINALIZE = False
    torch.cuda.current_stream().wait_stream(side)
    xs.grad = None; ws.grad = None; bs.grad = None
    graph = torch.cuda.CUDAGraph()
    ON.FUSED_FINALIZE = True
    try:
        with torch.cuda.graph(graph):
            y = ON.instance_norm_act(xs, ws, bs, 1e-5, 0.01, None)
            y.backward(gy)
    finally:
        ON.FUSED_FINALIZE = False
    for _ in range(3):
        xs.grad.zero_(); ws.grad.zero_(); bs.grad.zero_()
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y.detach(), eager[0]) and torch.equal(xs.grad, eager[1])
    assert torch.equal(ws.grad, eager[2]) and torch.equal(bs.grad, eager[3])
