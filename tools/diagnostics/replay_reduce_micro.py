"""Pure PyTorch: capture -> replay -> eager -> replay of a backward with a broadcast-added bias; where does the replay's bias gradient go?"""
import torch
dev = "cuda"; bf = torch.bfloat16
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
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
torch.cuda.synchronize()
def show(tag):
    torch.cuda.synchronize(); print(f"{tag:40s} b.grad ptr {b.grad.data_ptr():#x} |b.grad| {float(b.grad.norm()):.6g}  |w.grad| {float(w.grad.norm()):.6g}")
b.grad.zero_(); w.grad.zero_(); step(); show("eager reference")
ref = b.grad.clone()
b.grad.zero_(); w.grad.zero_(); g.replay(); show("replay 1")
b.grad.zero_(); w.grad.zero_(); g.replay(); show("replay 2")
b.grad.zero_(); w.grad.zero_(); step(); show("eager")
b.grad.zero_(); w.grad.zero_(); g.replay(); show("replay 3 (after eager)")
print("replay 3 vs eager, elementwise ratio (first 8):", (b.grad / ref)[:8].tolist())
b.grad.zero_(); w.grad.zero_(); g.replay(); g.replay(); show("two replays accumulated")
