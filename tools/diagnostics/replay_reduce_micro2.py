"""Which eager operation between two replays breaks the captured column sum?  Pure PyTorch."""
import torch
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(2048, 64, device=dev)
out = torch.zeros(64, device=dev)
def fn():
    out.copy_((x * 1.5).sum(dim=0))
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): fn()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fn()
ref = (x.double() * 1.5).sum(dim=0).float()
def replay(tag):
    out.zero_(); g.replay(); torch.cuda.synchronize()
    print(f"{tag:70s} rel err {float((out - ref).norm() / ref.norm()):.3g}")
replay("first replay")
replay("second replay (nothing in between)")
y = torch.randn(1000, device=dev).sum(); torch.cuda.synchronize()
replay("after an unrelated small eager reduction")
z = torch.randn(4096, 64, device=dev).sum(dim=0); torch.cuda.synchronize()
replay("after an eager column sum of another tensor (4096 x 64)")
z = (x * 1.5).sum(dim=0); torch.cuda.synchronize()
replay("after the same eager column sum")
replay("and once more")
a = torch.randn(512, 512, device=dev); b = a @ a; torch.cuda.synchronize()
replay("after an eager GEMM")
