"""PMC target (round 4): the fused backward of the 1x1 output convolution (csrc/out_conv_bwd.hip) at 2 x 512 x 512 x 64 bf16, 60 real
output channels, 10 launches.  Run under `rocprofv3 --pmc <group> --kernel-trace --output-format csv`, one counter group per pass."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py4cast_amd import _lib as L
from py4cast_amd import ops_model as om
dev = torch.device("cuda:0")
B, N, CO = 2, 512 * 512, 60
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
dy = rn(B, N, 64)
dy[..., CO:] = 0
dy, y = dy.bfloat16(), rn(B, N, 64).bfloat16()
w = rn(CO, 64) * 0.2
sc, sh, mu, rs = torch.rand(B, 64, device=dev) + 0.5, rn(B, 64) * 0.3, rn(B, 64) * 0.2, torch.rand(B, 64, device=dev) + 0.5
gw = torch.zeros(CO, 64, device=dev)
for _ in range(10):
    om.out_conv_bwd(dy, w, y, sc, sh, mu, rs, gw)
torch.cuda.synchronize()
print("done")
