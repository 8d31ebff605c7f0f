import sys, torch
sys.path.insert(0, '/root/repo')
from py4cast_amd import ops
dev = torch.device('cuda:0')
B, T, H, W, F = 2, 3, 512, 512, 60
def t(fn, n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn()
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1000
x = torch.randn(B, T, H, W, F, device=dev); y = torch.randn(B, T, H, W, F, device=dev)
std = torch.rand(F, device=dev) + 0.5; mean = torch.randn(F, device=dev)
raw = torch.randn(F, B, T + 1, H, W, device=dev)
nb = x.numel() * 4
us = t(lambda: ops.unnormalize(x, std, mean, out=x)); print("unnormalize (B,T,512,512,60): %.1f us, %.2f TB/s" % (us, 2 * nb / us / 1e6))
us = t(lambda: ops.acc_sums(x, y, ops.MaskSpec(0), mean)); print("acc_sums: %.1f us, %.2f TB/s" % (us, 2 * nb / us / 1e6))
us = t(lambda: ops.pack_standardize(raw, mean, std)); print("pack_standardize (60 planes x B x 4 steps): %.1f us, %.2f TB/s" % (us, 2 * raw.numel() * 4 / us / 1e6))
