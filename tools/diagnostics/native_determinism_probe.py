"""Bitwise run-to-run determinism of the native autograd nodes at the 512 x 512 bench sizes (forward + every gradient, 3 runs)."""
import sys, torch
sys.path.insert(0, ".")
import py4cast_amd.ops_model as OM, py4cast_amd.ops_inorm as ON, py4cast_amd.ops_rows as R, py4cast_amd.ops_ts as TS
from py4cast_amd.ops_attention import window_attention
dev = "cuda"
def check(name, fn, inputs, n=3):
    outs = []
    for _ in range(n):
        ins = [t.detach().clone().requires_grad_(t.is_floating_point()) for t in inputs]
        y = fn(*ins)
        g = torch.autograd.grad(y, [t for t in ins if t.requires_grad], torch.ones_like(y) * 0.37)
        outs.append([y.detach()] + [t.detach() for t in g])
    same = all(torch.equal(a, b) for k in range(1, n) for a, b in zip(outs[0], outs[k]))
    worst = max(float((a.float() - b.float()).abs().max() / a.float().abs().max().clamp_min(1e-30)) for k in range(1, n) for a, b in zip(outs[0], outs[k]))
    print(f"{name:60s} {'bit-identical' if same else 'DIFFERS'}  worst {worst:.2g}")
torch.manual_seed(0)
bf = torch.bfloat16
x96 = torch.randn(2, 512, 512, 96, device=dev, dtype=bf); x48 = torch.randn(2, 512, 512, 48, device=dev, dtype=bf)
w = torch.randn(48, 96, 3, 3, device=dev) * 0.05; w1 = torch.randn(48, 96, 1, 1, device=dev) * 0.05; w2 = torch.randn(48, 48, 3, 3, device=dev) * 0.05
check("conv_nhwc 96->48 3x3 (2,512,512)", lambda x, w: OM.conv_nhwc(x, w), [x96, w])
check("conv_nhwc 96->48 1x1", lambda x, w: OM.conv_nhwc(x, w), [x96, w1])
check("conv_nhwc 48->48 3x3", lambda x, w: OM.conv_nhwc(x, w), [x48, w2])
x64 = torch.randn(2, 512, 512, 64, device=dev, dtype=bf); w64 = torch.randn(64, 64, 3, 3, device=dev) * 0.05
check("conv_nhwc 64->64 3x3", lambda x, w: OM.conv_nhwc(x, w), [x64, w64])
g48, b48 = torch.rand(48, device=dev) + 0.5, torch.randn(48, device=dev)
check("instance_norm_act (2,512,512,48) + res", lambda x, g, b, r: ON.instance_norm_act(x, g, b, 1e-5, 0.01, r), [x48, g48, b48, torch.randn_like(x48)])
x128 = torch.randn(2, 128, 128, 192, device=dev, dtype=bf)
check("instance_norm_act (2,128,128,192)", lambda x, g, b: ON.instance_norm_act(x, g, b, 1e-5, 0.01, None), [x128, torch.rand(192, device=dev) + 0.5, torch.randn(192, device=dev)])
for C, Rr in ((48, 2 * 256 * 256), (96, 2 * 128 * 128), (192, 2 * 64 * 64), (384, 2 * 32 * 32), (128, 2 * 128 * 128), (256, 2 * 64 * 64), (512, 2 * 32 * 32)):
    xr = torch.randn(Rr, C, device=dev, dtype=bf)
    check(f"row_layer_norm R={Rr} C={C}", lambda x, g, b: R.row_layer_norm(x, g, b, 1e-5), [xr, torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)])
for C, heads, hw in ((48, 3, 256), (96, 6, 128), (192, 12, 64), (384, 24, 32)):
    qkv = torch.randn(2, hw + (-hw) % 7, hw + (-hw) % 7, 3 * C, device=dev, dtype=bf)
    bias = torch.randn(heads, 49, 49, device=dev)
    for shift in (0, 3):
        check(f"window_attention C={C} heads={heads} grid={qkv.shape[1]} shift={shift}", lambda q, b: window_attention(q, b, heads, 7, shift), [qkv, bias])
xl = torch.randn(2, 256, 256, 48, device=dev, dtype=bf); wl = torch.randn(192, 48, device=dev) * 0.1; bl = torch.randn(192, device=dev)
check("linear_nd (2,256,256,48)->192 [library GEMMs]", lambda x, w, b: R.linear_nd(x, w, b), [xl, wl, bl])
N, d, h = 128 * 128, 8, 16
big = torch.randn(2, N, 4, h, d, device=dev, dtype=bf)
q, k = big[:, :, 0].permute(0, 2, 1, 3), big[:, :, 1].permute(0, 2, 1, 3)
check("ts.gram N=16384 d=8", lambda a: TS.gram(a[:, :, 0].permute(0, 2, 1, 3), a[:, :, 1].permute(0, 2, 1, 3)), [big])
M = torch.randn(2, h, d, 64, device=dev)
check("ts.apply N=16384 d=8 -> 64", lambda a, m: TS.apply(a[:, :, 0].permute(0, 2, 1, 3), m), [big, M])
