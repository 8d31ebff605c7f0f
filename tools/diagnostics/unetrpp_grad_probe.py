"""Capture the input / output gradient of decoder2's residual block inside the model, then replay the block alone on the captured
tensors: GPU (native) and CPU fp64."""
import sys, copy, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_unetrpp_gpu import _pair, _rel
H, W, cin, cout = 64, 96, 13, 5
model, oracle = _pair(cin, cout, (H, W), linear=False)
model = model.cuda().train()
blk = model.decoder2.decoder_block[0]
cap = {}
blk.register_forward_hook(lambda m, i, o: cap.update(x=i[0].detach().clone(), xs=i[0].stride(), y=o.detach().clone()))
blk.register_full_backward_hook(lambda m, gi, go: cap.update(gin=gi[0].detach().clone(), gout=go[0].detach().clone(), gs=go[0].stride()))
torch.manual_seed(42)
x, gy = torch.randn(2, H, W, cin), torch.randn(2, H, W, cout)
xg = x.cuda().requires_grad_(True)
model(xg).backward(gy.cuda())
print("input strides", cap["xs"], "grad_out strides", cap["gs"])
ref = copy.deepcopy(blk).double().cpu().train()
ref._forward_hooks.clear(); ref._backward_hooks.clear()
xb = cap["x"].double().cpu().requires_grad_()
yb = ref(xb); gb, = torch.autograd.grad(yb, [xb], cap["gout"].double().cpu())
print("in-model forward vs fp64 replay", _rel(cap["y"], yb), " in-model grad_in vs fp64 replay", _rel(cap["gin"], gb))
blk2 = copy.deepcopy(blk); blk2._forward_hooks.clear(); blk2._backward_hooks.clear()
xa = cap["x"].clone().requires_grad_()
ya = blk2(xa); ga, = torch.autograd.grad(ya, [xa], cap["gout"])
print("GPU replay forward", _rel(ya, yb), "grad_in", _rel(ga, gb))

# ---- step through the block on the captured values
import torch.nn.functional as F
import py4cast_amd.unetrpp as U
import py4cast_amd.ops_inorm as ON
def steps(b, x, g, native):
    if native:
        inorm = lambda m, t, slope=1.0, res=None: ON.instance_norm_act(t.permute(0, 2, 3, 1), m.weight, m.bias, m.eps, slope, None if res is None else res.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        y1 = U._conv(b.conv1, x).contiguous(memory_format=torch.channels_last); y1.retain_grad()
        n1 = inorm(b.norm1, y1, 0.01); n1.retain_grad()
        y2 = U._conv(b.conv2, n1).contiguous(memory_format=torch.channels_last); y2.retain_grad()
        out = inorm(b.norm2, y2, 0.01, x)
    else:
        y1 = b.conv1(x); y1.retain_grad()
        n1 = F.leaky_relu(b.norm1(y1), 0.01); n1.retain_grad()
        y2 = b.conv2(n1); y2.retain_grad()
        out = F.leaky_relu(b.norm2(y2) + x, 0.01)
    out.backward(g)
    return {"y1": y1, "n1": n1, "y2": y2, "out": out, "d_y2": y2.grad, "d_n1": n1.grad, "d_y1": y1.grad, "d_x": x.grad}
xa = cap["x"].clone().requires_grad_()
A = steps(blk2, xa, cap["gout"], True)
xb = cap["x"].double().cpu().requires_grad_()
Bv = steps(ref, xb, cap["gout"].double().cpu(), False)
for k in A:
    print(k, _rel(A[k], Bv[k]), "max|ref|", float(Bv[k].abs().max()))
y2 = Bv["y2"].detach()
print("conv2 output per (sample, channel): mean / std", (y2.mean(dim=(2, 3)).abs() / y2.std(dim=(2, 3))).max().item(), "min std", y2.std(dim=(2, 3)).min().item())
y1 = Bv["y1"].detach()
print("conv1 output per (sample, channel): |mean| / std max", (y1.mean(dim=(2, 3)).abs() / y1.std(dim=(2, 3))).max().item(), "min std", y1.std(dim=(2, 3)).min().item())

d = (A["d_y2"].detach().double().cpu() - Bv["d_y2"].detach()).abs()
print("elements of d_y2 off by > 1e-4:", int((d > 1e-4).sum()), "of", d.numel(), " > 1e-5:", int((d > 1e-5).sum()), " > 1e-6:", int((d > 1e-6).sum()))
idx = (d > 1e-4).nonzero()[:8]
with torch.no_grad():
    z = ref.norm2(Bv["y2"].detach()) + xb.detach()
for i in idx:
    i = tuple(int(v) for v in i)
    print("  at", i, "pre-activation (fp64)", float(z[i]), "gpu d_y2", float(A["d_y2"][i]), "ref", float(Bv["d_y2"][i]), "gout", float(cap["gout"][i]))
# per-channel structure of the error
print("max error per channel:", [float(f"{v:.1e}") for v in d.amax(dim=(0, 2, 3))])
print("max error per sample:", [float(f"{v:.1e}") for v in d.amax(dim=(1, 2, 3))])
j = tuple(int(v) for v in (d == d.max()).nonzero()[0])
za = blk2.norm2(A["y2"].detach()) + xa.detach()
print("largest error at", j, "pre-activation fp64", float(z[j]), "fp32 (GPU)", float(za[j]), "gout", float(cap["gout"][j]))
