"""PMC target (round 4): the row-streaming weight-gradient kernel at 2 x 512 x 512 x 64 bf16, 128 workgroups (the plan's geometry: 8
segments per strip) and 256 (16 segments), plain / input transform / NormBwdCoef, 10 launches each.  Run under
`rocprofv3 --pmc <group> --kernel-trace --output-format csv`, one counter group per pass."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py4cast_amd import _lib as L
dev = torch.device("cuda:0")
B, H, W = 2, 512, 512
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
x, dA, y = rn(B, H, W, 64).bfloat16(), rn(B, H, W, 64).bfloat16(), rn(B, H, W, 64).bfloat16()
sc, sh = torch.rand(B, 64, device=dev) + 0.5, rn(B, 64) * 0.1
gamma, nsc, nsh = torch.rand(64, device=dev) + 0.5, torch.rand(B, 64, device=dev) + 0.5, rn(B, 64) * 0.3
rstd, mean, k1, k2 = torch.rand(B, 64, device=dev) + 0.5, rn(B, 64) * 0.2, rn(B, 64) * 0.1, rn(B, 64) * 0.1
grad = torch.zeros(64, 64, 3, 3, device=dev)
ws = torch.empty(L.lib().p4c_conv_wgrad_workspace_bytes(64, 3) // 4, dtype=torch.float32, device=dev)
st = L.stream(dev)
for nseg in ("8", "16"):
    os.environ["P4C_WGROWS_NSEG"] = nseg
    for transform, nb in ((False, False), (True, False), (True, True)):
        a = (L.ptr(sc), L.ptr(sh), 1) if transform else (None, None, 0)
        for _ in range(10):
            if nb:
                L.call("p4c_conv_wgrad_nb", L.ptr(x), a[0], a[1], a[2], L.ptr(dA), L.ptr(y), L.ptr(gamma), L.ptr(nsc), L.ptr(nsh), L.ptr(rstd),
                       L.ptr(mean), L.ptr(k1), L.ptr(k2), 64, 64, L.ptr(grad), L.ptr(ws), B, H, W, st)
            else:
                L.call("p4c_conv_wgrad", L.ptr(x), L.BF16, L.BF16, 64, 3, a[0], a[1], a[2], L.ptr(dA), 64, 64, L.ptr(grad), L.ptr(ws), B, H, W, st)
        torch.cuda.synchronize()
print("done")
