"""A few dozen launches of the implicit-GEMM convolution (csrc/gemm.hip) at UNETR++'s stage-0 shape (2 x 128 x 128, 128 -> 128) and its
stage-3 shape (2 x 16 x 16, 1024 -> 1024: split-K) for the rocprofv3 --pmc passes of tools/diagnostics/r05_profile.sh."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from py4cast_amd import ops_gemm as G

dev = torch.device("cuda:0")
for (B, H, W, C) in [(2, 128, 128, 128), (2, 16, 16, 1024)]:
    x = torch.randn(B, H, W, C, device=dev).bfloat16().view(-1, C)
    dy = torch.randn(B, H, W, C, device=dev).bfloat16().view(-1, C)
    w = torch.randn(C, C, 3, 3, device=dev) / (9 * C) ** 0.5
    fwd, dgr = G.weight_images(w, 9)
    for _ in range(20):
        G.gemm_nt(x, fwd, C, 9 * C, conv=(H, W, C), want_stats=True)
        G.gemm_tn(dy, x, C, C, conv=(H, W))
torch.cuda.synchronize()
