"""Micro-reproducer attempt: the broadcast-add bias gradient (autograd sum_to_size) in a captured backward, replayed after an eager pass."""
import torch
dev = "cuda"
torch.manual_seed(0)
def run_case(shape, dtype_in, flat_view):
    b = torch.randn(shape[-1], device=dev, requires_grad=True)
    w = torch.randn(shape[-1], shape[-1], device=dev, requires_grad=True)
    if flat_view:
        flat = torch.zeros(b.numel() + w.numel(), device=dev)
        b.grad = flat[: b.numel()].view_as(b); w.grad = flat[b.numel():].view_as(w)
    else:
        b.grad = torch.zeros_like(b); w.grad = torch.zeros_like(w)
    x = torch.randn(*shape, device=dev).to(dtype_in)
    def step():
        h = (x @ w.to(dtype_in)).float() + b
        (h.sin() * 1e-3).sum().backward()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    def grads(fn):
        b.grad.zero_(); w.grad.zero_(); fn(); torch.cuda.synchronize(); return b.grad.clone(), w.grad.clone()
    r0 = grads(g.replay); e0 = grads(step); r1 = grads(g.replay); r2 = grads(g.replay)
    rel = lambda a, c: float((a - c).norm() / c.norm())
    print(shape, dtype_in, "flat" if flat_view else "own", "| replay0 vs eager: b %.2g w %.2g | replay after eager vs eager: b %.2g w %.2g | next replay: b %.2g" % (
        rel(r0[0], e0[0]), rel(r0[1], e0[1]), rel(r1[0], e0[0]), rel(r1[1], e0[1]), rel(r2[0], e0[0])))
for shape in ((2, 2, 512, 64), (2, 2, 1024, 64), (2, 2, 128, 64), (2, 16384, 64), (524288, 64), (2, 2, 512, 32)):
    for dt in (torch.bfloat16, torch.float32):
        for fv in (True, False):
            run_case(shape, dt, fv)
