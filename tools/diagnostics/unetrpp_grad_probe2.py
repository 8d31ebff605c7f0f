"""Is the 1e-3 gradient error of the linear_upsampling=False configuration the product's or fp32 conditioning?  The oracle itself
in fp32 on the GPU (library ops only), and in fp32 on the CPU, against the oracle in fp64."""
import sys, copy, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_unetrpp_gpu import _pair, _rel
H, W, cin, cout = 64, 96, 13, 5
model, oracle = _pair(cin, cout, (H, W), linear=False)
oracle.train()
torch.manual_seed(42)
x, gy = torch.randn(2, H, W, cin), torch.randn(2, H, W, cout)
xr = x.double().requires_grad_(True)
yr = oracle(xr); yr.backward(gy.double())
ref = dict(oracle.named_parameters())
for dev, cudnn in (("cpu", True), ("cuda", True), ("cuda", False)):
    torch.backends.cudnn.enabled = cudnn
    o32 = copy.deepcopy(oracle).float().to(dev).train()
    for p in o32.parameters():
        p.grad = None
    x32 = x.clone().to(dev).requires_grad_(True)
    x32.retain_grad()
    y32 = o32(x32); y32.backward(gy.to(dev))
    errs = sorted(((_rel(p.grad, ref[n].grad), n) for n, p in o32.named_parameters() if p.grad is not None), reverse=True)
    print(dev, "cudnn", cudnn, "y", _rel(y32, yr), "dx", None if x32.grad is None else _rel(x32.grad, xr.grad), errs[:3], sum(p.grad is None for p in o32.parameters()))
