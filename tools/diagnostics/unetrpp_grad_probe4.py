"""Which native node carries the gradient error of the linear_upsampling=False configuration: switch them off one at a time."""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_unetrpp_gpu import _pair, _rel
import py4cast_amd.unetrpp as U
import py4cast_amd.ops_inorm as ON
import py4cast_amd.ops_model as OM
import py4cast_amd.ops_rows as R
import torch.nn.functional as F
H, W, cin, cout = 64, 96, 13, 5
on_sup, om_sup, lin = ON.supported, OM.conv_nhwc_supported, R.linear_nd
def run(tag):
    model, oracle = _pair(cin, cout, (H, W), linear=False)
    model = model.cuda().train(); oracle.train()
    torch.manual_seed(42)
    x, gy = torch.randn(2, H, W, cin), torch.randn(2, H, W, cout)
    xg = x.cuda().requires_grad_(True)
    model(xg).backward(gy.cuda())
    xr = x.double().requires_grad_(True)
    oracle(xr).backward(gy.double())
    ref = dict(oracle.named_parameters())
    worst = max((_rel(p.grad, ref[n].grad), n) for n, p in model.named_parameters())
    print(tag, "dx", _rel(xg.grad, xr.grad), worst)
run("default")
ON.supported = lambda x: False
run("no native instance norm")
ON.supported = on_sup
OM.conv_nhwc_supported = lambda x, w: False
run("no native conv")
OM.conv_nhwc_supported = om_sup
R.linear_nd = lambda x, w, b=None: F.linear(x, w.to(x.dtype), None if b is None else b.to(x.dtype))
run("library linear")
