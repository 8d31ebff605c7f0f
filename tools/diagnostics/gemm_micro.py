"""Times the wide-channel GEMM / convolution kernels (csrc/gemm.hip) at the UNETR++ / SwinUNETR bench shapes next to the library
calls they replace (torch F.linear / F.conv2d channels_last, bf16).  Usage: python tools/diagnostics/gemm_micro.py [--quick]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from py4cast_amd import ops_gemm as G

dev = torch.device("cuda:0")


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us


def conv_case(B, H, W, Ci, Co):
    x = torch.randn(B, H, W, Ci, device=dev).bfloat16()
    w = torch.randn(Co, Ci, 3, 3, device=dev) / (9 * Ci) ** 0.5
    dy = torch.randn(B, H, W, Co, device=dev).bfloat16()
    fwd, dgr = G.weight_images(w, 9)
    xm, dym = x.view(-1, Ci), dy.view(-1, Co)
    t_f = timeit(lambda: G.gemm_nt(xm, fwd, Co, 9 * Ci, conv=(H, W, Ci), want_stats=True))
    t_d = timeit(lambda: G.gemm_nt(dym, dgr, Ci, 9 * Co, conv=(H, W, Co)))
    t_w = timeit(lambda: G.gemm_tn(dym, xm, Co, Ci, conv=(H, W)))
    xl = x.permute(0, 3, 1, 2)
    wl = w.bfloat16().contiguous(memory_format=torch.channels_last)
    t_lib = timeit(lambda: F.conv2d(xl, wl, padding=1))
    gf = 2 * 9 * Ci * Co * B * H * W / 1e9
    print(f"conv3x3 B{B} {H}x{W} {Ci}->{Co}: fwd {t_f:7.1f} us ({gf / t_f * 1e3:6.0f} TF/s)  dgrad {t_d:7.1f}  wgrad {t_w:7.1f}  | library fwd {t_lib:7.1f}", flush=True)


def lin_case(R, K, N):
    x = torch.randn(R, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev) / K ** 0.5
    dy = torch.randn(R, N, device=dev).bfloat16()
    fwd, dgr = G.weight_images(w, 1)
    t_f = timeit(lambda: G.gemm_nt(x, fwd, N, K))
    t_d = timeit(lambda: G.gemm_nt(dy, dgr, K, N))
    t_w = timeit(lambda: G.gemm_tn(dy, x, N, K, want_bias=True))
    wq = w.bfloat16()
    t_lib = timeit(lambda: F.linear(x, wq))
    t_libw = timeit(lambda: dy.t() @ x)
    gf = 2 * R * K * N / 1e9
    print(f"linear R{R} {K}->{N}: fwd {t_f:7.1f} us ({gf / t_f * 1e3:6.0f} TF/s)  dgrad {t_d:7.1f}  wgrad+db {t_w:7.1f}  | library fwd {t_lib:7.1f} wgrad {t_libw:7.1f}", flush=True)


if __name__ == "__main__":
    for (B, H, W, C) in [(2, 128, 128, 128), (2, 64, 64, 256), (2, 32, 32, 512), (2, 16, 16, 1024)]:
        conv_case(B, H, W, C, C)
    conv_case(2, 32, 32, 384, 384)      # SwinUNETR encoder10 at 512 x 512 / 16... (16 fs = 384)
    conv_case(2, 64, 64, 192, 96)
    for (R, K, N) in [(32768, 128, 512), (8192, 256, 1024), (2048, 512, 2048), (512, 1024, 4096), (32768, 128, 64), (512, 1024, 512),
                      (8192, 96, 288), (8192, 96, 384), (8192, 384, 96), (2048, 192, 768), (512, 384, 1536)]:
        lin_case(R, K, N)
