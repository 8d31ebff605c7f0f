import sys, torch
sys.path.insert(0, ".")
import py4cast_amd
from py4cast_amd.ops_inorm import group_norm
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, W, C, G) in ((2, 128, 128, 128, 1), (2, 64, 64, 256, 128), (2, 32, 32, 512, 256), (2, 16, 16, 1024, 512)):
    x = torch.randn(B, H, W, C, device="cuda").bfloat16().requires_grad_(True)
    g = torch.rand(C, device="cuda", requires_grad=True); b = torch.randn(C, device="cuda", requires_grad=True)
    dy = torch.randn_like(x)
    def nat():
        y = group_norm(x, G, g, b); y.backward(dy)
    m = torch.nn.GroupNorm(G, C).cuda()
    xn = x.detach().permute(0, 3, 1, 2).requires_grad_(True)
    def lib():
        y = m(xn.float()).to(torch.bfloat16); y.backward(dy.permute(0, 3, 1, 2))
    print((B, H, W, C, G), "native fwd+bwd %.1f us" % t(nat), " library %.1f us" % t(lib))
