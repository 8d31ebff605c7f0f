"""Per-parameter gradient error of UNetRPPMI355X against the fp64 oracle (the configuration of tests/test_unetrpp_gpu.py)."""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_unetrpp_gpu import _pair, _rel
for linear in (False, True):
    for rep in range(2):
        H, W, cin, cout = 64, 96, 13, 5
        model, oracle = _pair(cin, cout, (H, W), linear=linear)
        model = model.cuda().train(); oracle.train()
        torch.manual_seed(42)
        x, gy = torch.randn(2, H, W, cin), torch.randn(2, H, W, cout)
        xg = x.cuda().requires_grad_(True)
        y = model(xg); y.backward(gy.cuda())
        xr = x.double().requires_grad_(True)
        yr = oracle(xr); yr.backward(gy.double())
        ref = dict(oracle.named_parameters())
        errs = sorted(((_rel(p.grad, ref[n].grad), n, float(ref[n].grad.abs().max())) for n, p in model.named_parameters()), reverse=True)
        print("linear", linear, "rep", rep, "y", _rel(y, yr), "dx", _rel(xg.grad, xr.grad))
        for e in errs[:6]:
            print("   ", e)
