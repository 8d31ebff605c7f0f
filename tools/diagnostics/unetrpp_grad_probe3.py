"""Module-by-module comparison of the gradients arriving at / leaving every container module (product on the GPU vs fp64 oracle)."""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_unetrpp_gpu import _pair, _rel
H, W, cin, cout = 64, 96, 13, 5
model, oracle = _pair(cin, cout, (H, W), linear=False)
model = model.cuda().train(); oracle.train()
order = []
def hook(store, name, rec_order):
    def f(mod, gin, gout):
        store[name] = (gin[0].detach().double().cpu() if gin and gin[0] is not None else None, gout[0].detach().double().cpu() if gout[0] is not None else None)
        if rec_order:
            order.append(name)
    return f
gp, go = {}, {}
for n, m in model.named_modules():
    if n:
        m.register_full_backward_hook(hook(gp, n, True))
for n, m in oracle.named_modules():
    if n:
        m.register_full_backward_hook(hook(go, n, False))
torch.manual_seed(42)
x, gy = torch.randn(2, H, W, cin), torch.randn(2, H, W, cout)
xg = x.cuda().requires_grad_(True)
model(xg).backward(gy.cuda())
xr = x.double().requires_grad_(True)
oracle(xr).backward(gy.double())
def cmp(a, b):
    if a is None or b is None:
        return None
    if a.shape != b.shape:
        if a.numel() == b.numel() and a.dim() == 4 and b.dim() == 4 and a.permute(0, 3, 1, 2).shape == b.shape:
            a = a.permute(0, 3, 1, 2)
        elif a.numel() == b.numel() and a.dim() == 4 and b.dim() == 4 and a.permute(0, 2, 3, 1).shape == b.shape:
            a = a.permute(0, 2, 3, 1)
        else:
            return f"shape {tuple(a.shape)} vs {tuple(b.shape)}"
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for n in order:
    if n in go:
        e_in, e_out = cmp(gp[n][0], go[n][0]), cmp(gp[n][1], go[n][1])
        print(f"{n:50s} grad_out {e_out}   grad_in {e_in}")
