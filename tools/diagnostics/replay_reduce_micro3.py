"""Variants of the micro-reproducer (captured backward with a broadcast-added bias), pure PyTorch."""
import torch
dev = "cuda"
def case(tag, make_h, between):
    torch.manual_seed(0)
    b = torch.randn(64, device=dev, requires_grad=True); w = torch.randn(64, 64, device=dev, requires_grad=True)
    b.grad = torch.zeros_like(b); w.grad = torch.zeros_like(w)
    x = torch.randn(2, 2, 512, 64, device=dev)
    def step():
        (make_h(x, w, b).sin() * 1e-3).sum().backward()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    def grads(fn):
        b.grad.zero_(); w.grad.zero_(); fn(); torch.cuda.synchronize(); return b.grad.clone()
    e = grads(step)            # reference (eager)
    r0 = grads(g.replay)
    between(step, x, w, b)
    torch.cuda.synchronize()
    r1 = grads(g.replay)
    rel = lambda a, c: float((a - c).norm() / c.norm())
    print(f"{tag:75s} replay0 {rel(r0, e):.2g}   replay1 {rel(r1, e):.2g}")
nothing = lambda step, x, w, b: None
def eager_step(step, x, w, b): step()
def eager_fwd(step, x, w, b):
    with torch.no_grad(): (x @ w + b).sum()
def eager_bwd_other(step, x, w, b):
    q = torch.randn(64, device=dev, requires_grad=True); ((x + q).sin().sum()).backward()
bf = torch.bfloat16
# note: here the reference eager pass `e` already runs BEFORE replay0
case("bf16 matmul .float() + b | nothing between", lambda x, w, b: (x.to(bf) @ w.to(bf)).float() + b, nothing)
case("bf16 matmul .float() + b | eager step between", lambda x, w, b: (x.to(bf) @ w.to(bf)).float() + b, eager_step)
case("fp32 matmul + b          | eager step between", lambda x, w, b: x @ w + b, eager_step)
case("x + b (no matmul)        | eager step between", lambda x, w, b: x * w[0] + b, eager_step)
case("fp32 matmul + b          | eager forward only between", lambda x, w, b: x @ w + b, eager_fwd)
case("fp32 matmul + b          | eager backward of ANOTHER broadcast add between", lambda x, w, b: x @ w + b, eager_bwd_other)
