"""The micro-reproducer (captured backward with a broadcast-added bias) with and without p4c_graph_replace_memsets."""
import ctypes, sys, torch
sys.path.insert(0, ".")
from py4cast_amd import _lib as L
dev = "cuda"; bf = torch.bfloat16
def case(fix):
    torch.manual_seed(0)
    b = torch.randn(64, device=dev, requires_grad=True); w = torch.randn(64, 64, device=dev, requires_grad=True)
    b.grad = torch.zeros_like(b); w.grad = torch.zeros_like(w)
    x = torch.randn(2, 2, 512, 64, device=dev)
    def step():
        h = (x.to(bf) @ w.to(bf)).float() + b
        (h.sin() * 1e-3).sum().backward()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        step()
    info = ""
    if fix:
        rep, left = ctypes.c_int(), ctypes.c_int()
        L.check(L.lib().p4c_graph_replace_memsets(ctypes.c_void_p(g.raw_cuda_graph()), ctypes.byref(rep), ctypes.byref(left)), "p4c_graph_replace_memsets")
        info = f"(memset nodes replaced {rep.value}, left {left.value})"
    g.instantiate()
    def grads(fn):
        b.grad.zero_(); w.grad.zero_(); fn(); torch.cuda.synchronize(); return b.grad.clone(), w.grad.clone()
    e = grads(step)
    rel = lambda a, c: float((a - c).norm() / c.norm())
    out = []
    for i in range(4):
        r = grads(g.replay); out.append((round(rel(r[0], e[0]), 4), round(rel(r[1], e[1]), 4)))
    print("fix" if fix else "no fix", info, "replays 1..4 (bias err, weight err):", out)
case(False)
case(True)
