"""Check every native convolution node of the linear_upsampling=False model in place: its forward, data and weight gradient against
the library convolution in fp64 on the same tensors."""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_unetrpp_gpu import _pair, _rel
import py4cast_amd.ops_model as OM
import torch.nn.functional as F
H, W, cin, cout = 64, 96, 13, 5
orig_bwd = OM._ConvNHWC.backward
def bwd(ctx, dy):
    x, w = ctx.saved_tensors
    out = orig_bwd(ctx, dy)
    dx, dw = out[0], out[1]
    xd = x.detach().double().permute(0, 3, 1, 2).requires_grad_(True)
    wd = torch.zeros(w.shape[0], x.shape[-1], *w.shape[2:], dtype=torch.float64, device=w.device)
    wd[:, : w.shape[1]] = w.detach().double()
    wd.requires_grad_(True)
    with torch.enable_grad():
        y = F.conv2d(xd, wd, None, padding=w.shape[-1] // 2)
        gx, gw = torch.autograd.grad(y, [xd, wd], dy.double().permute(0, 3, 1, 2))
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))
    print("conv", tuple(x.shape), tuple(w.shape), "dy contiguous", dy.is_contiguous(), tuple(dy.stride()),
          "dx", None if dx is None else rel(dx.permute(0, 3, 1, 2), gx), "dw", None if dw is None else rel(dw, gw[:, : w.shape[1]]))
    return out
OM._ConvNHWC.backward = staticmethod(bwd)
model, oracle = _pair(cin, cout, (H, W), linear=False)
model = model.cuda().train()
torch.manual_seed(42)
x, gy = torch.randn(2, H, W, cin), torch.randn(2, H, W, cout)
xg = x.cuda().requires_grad_(True)
model(xg).backward(gy.cuda())
oracle.train()
xr = x.double().requires_grad_(True)
oracle(xr).backward(gy.double())
ref = dict(oracle.named_parameters())
worst = max((_rel(p.grad, ref[n].grad), n) for n, p in model.named_parameters())
print("with the in-place checks (each one synchronises): dx", _rel(xg.grad, xr.grad), worst)
