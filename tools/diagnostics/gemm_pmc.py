"""A few dozen launches of the implicit-GEMM convolution / tiled GEMM (csrc/gemm.hip) for the rocprofv3 --pmc passes of
tools/diagnostics/r06_profile.sh, on shapes whose launches differ in flavour / grid size so that pmc_summary.py can tell them apart:
  * conv3x3 128 -> 128 on 2 x 128 x 128 (UNETR++ stage 0)          gemm_nt<true>  grid 256 wg, gemm_tn<true, ..>
  * conv3x3 1024 -> 1024 on 1 x 16 x 16 (stage-3 widths, split-K 8)  gemm_nt<true>  grid 128 wg
  * a plain Linear 32 768 x 128 -> 128: every A row is read by exactly ONE tile through the direct-to-LDS loads -- a known-size copy
    through `buffer_load_dwordx4 ... lds` (8.39 MB in, 8.39 MB out, 32 KB of weights): what FETCH_SIZE counts for that instruction
    (is the gfx950 doubling needed?)                                 gemm_nt<false> grid 256 wg"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from py4cast_amd import ops_gemm as G

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "all"      # conv128 | conv1024 | linear | all (one shape per profiled process: the two
shapes = {"conv128": [(2, 128, 128, 128)], "conv1024": [(1, 16, 16, 1024)], "linear": [],   # convolutions launch the same grid size)
          "all": [(2, 128, 128, 128), (1, 16, 16, 1024)]}[which]
for (B, H, W, C) in shapes:
    x = torch.randn(B, H, W, C, device=dev).bfloat16().view(-1, C)
    dy = torch.randn(B, H, W, C, device=dev).bfloat16().view(-1, C)
    w = torch.randn(C, C, 3, 3, device=dev) / (9 * C) ** 0.5
    fwd, dgr = G.weight_images(w, 9)
    for _ in range(20):
        G.gemm_nt(x, fwd, C, 9 * C, conv=(H, W, C), want_stats=True)
        G.gemm_tn(dy, x, C, C, conv=(H, W))
if which in ("linear", "all"):
    xl = torch.randn(32768, 128, device=dev).bfloat16()
    wl = torch.randn(128, 128, device=dev) / 128 ** 0.5
    fwd, _ = G.weight_images(wl, 1)
    for _ in range(20):
        G.gemm_nt(xl, fwd, 128, 128)
torch.cuda.synchronize()
