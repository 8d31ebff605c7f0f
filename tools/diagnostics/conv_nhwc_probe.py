"""ops_model.conv_nhwc (padding / slicing glue included) against the library convolution in fp64."""
import sys, torch
sys.path.insert(0, ".")
import py4cast_amd.ops_model as OM
import torch.nn.functional as F
rel = lambda a, b: float((a.detach().double() - b.detach()).abs().max() / b.detach().abs().max().clamp_min(1e-30))
for (B, H, W, CI, CO, ks) in ((2, 64, 96, 16, 16, 3), (2, 64, 96, 13, 16, 3), (2, 64, 96, 13, 16, 1), (2, 16, 24, 32, 32, 3), (2, 8, 12, 64, 64, 3), (2, 32, 32, 96, 64, 3), (2, 32, 32, 40, 24, 3)):
    torch.manual_seed(0)
    x = torch.randn(B, H, W, CI, device="cuda", requires_grad=True)
    w = (torch.randn(CO, CI, ks, ks, device="cuda") * 0.1).requires_grad_()
    g = torch.randn(B, H, W, CO, device="cuda")
    y = OM.conv_nhwc(x, w)
    dx, dw = torch.autograd.grad(y, [x, w], g)
    xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    yr = F.conv2d(xd.permute(0, 3, 1, 2), wd, None, padding=ks // 2).permute(0, 2, 3, 1)
    gx, gw = torch.autograd.grad(yr, [xd, wd], g.double())
    print((B, H, W, CI, CO, ks), "y", rel(y, yr), "dx", rel(dx, gx), "dw", rel(dw, gw))
