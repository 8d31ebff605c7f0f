"""unetrpp.ResBlock on the GPU (native routes) against the same module in fp64 on the CPU (library ops only)."""
import sys, copy, torch
sys.path.insert(0, ".")
import py4cast_amd.unetrpp as U
rel = lambda a, b: float((a.detach().double().cpu() - b.detach()).abs().max() / b.detach().abs().max())
for cin, cout, H, W, norm, cl in ((16, 16, 64, 96, "instance", True), (16, 16, 64, 96, "instance", False), (32, 32, 16, 24, "batch", True), (32, 32, 16, 24, "batch", False),
                                  (64, 64, 8, 12, "batch", True), (13, 16, 64, 96, "instance", True)):
    torch.manual_seed(0)
    blk = U.ResBlock(cin, cout, norm)
    ref = copy.deepcopy(blk).double().train()
    blk = blk.cuda().train()
    x = torch.randn(2, cin, H, W)
    g = torch.randn(2, cout, H, W)
    xa = x.cuda()
    xa = (xa.contiguous(memory_format=torch.channels_last) if cl else xa).requires_grad_()
    xb = x.double().requires_grad_()
    ya = blk(xa); ga = torch.autograd.grad(ya, [xa] + list(blk.parameters()), g.cuda())
    yb = ref(xb); gb = torch.autograd.grad(yb, [xb] + list(ref.parameters()), g.double())
    names = ["x"] + [n for n, _ in blk.named_parameters()]
    print(cin, cout, H, W, norm, "channels_last" if cl else "nchw", "y", rel(ya, yb), {n: float(f"{rel(a, b):.2g}") for n, a, b in zip(names, ga, gb)})
