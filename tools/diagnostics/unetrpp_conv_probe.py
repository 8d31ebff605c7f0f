"""Compare unetrpp._conv / _conv_transpose (GEMM routes) against torch's convolutions on the GPU, fp32 and bf16."""
import torch, torch.nn as nn
import py4cast_amd.unetrpp as U
torch.manual_seed(0)
dev = "cuda"
cases = [("stem", nn.Conv2d(21, 128, 4, stride=4, bias=False), (2, 21, 64, 64)), ("down", nn.Conv2d(32, 64, 2, stride=2, bias=False), (2, 32, 16, 16)),
         ("conv8", nn.Conv2d(128, 128, 1), (2, 128, 16, 16)), ("out1", nn.Conv2d(64, 21, 1), (2, 64, 64, 64))]
for dt in (torch.float32, torch.bfloat16):
    for name, m, shp in cases:
        m = m.to(dev)
        x = torch.randn(*shp, device=dev).to(dt).contiguous(memory_format=torch.channels_last).requires_grad_()
        y = U._conv(m, x)
        yr = m._conv_forward(x.float(), m.weight, m.bias)
        g = torch.randn_like(yr)
        ps = [x, m.weight] + ([m.bias] if m.bias is not None else [])
        got = torch.autograd.grad(y, ps, g.to(dt)); ref = torch.autograd.grad(yr, ps, g)
        rel = lambda a, b: ((a.float() - b.float()).abs().max() / b.float().abs().max()).item()
        print(dt, name, "y", rel(y, yr), [rel(a, b) for a, b in zip(got, ref)])
    m = nn.ConvTranspose2d(64, 32, 2, stride=2, bias=False).to(dev)
    x = torch.randn(2, 64, 8, 8, device=dev).to(dt).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = U._conv_transpose(m, x); yr = nn.functional.conv_transpose2d(x.float(), m.weight, None, stride=2)
    g = torch.randn_like(yr)
    got = torch.autograd.grad(y, [x, m.weight], g.to(dt)); ref = torch.autograd.grad(yr, [x, m.weight], g)
    print(dt, "tconv", "y", rel(y, yr), [rel(a, b) for a, b in zip(got, ref)])
