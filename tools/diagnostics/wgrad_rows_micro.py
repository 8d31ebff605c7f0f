"""Round 4: the row-streaming weight-gradient kernel (csrc/conv_wgrad_rows.hip) against the tile kernel it replaces, alone on the chip,
launch + fixed-order reduction, at the benchmark's full resolution (2 x 512 x 512 x 64 bf16) and at the Titan grid (2 x 512 x 640):
plain / input transform / pass 2 of the normalisation backward on the way in (NormBwdCoef) / both.  One process, interleaved."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py4cast_amd import _lib as L
dev = torch.device("cuda:0")
def bench(B, H, W):
    g = torch.Generator(device=dev).manual_seed(1)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    x, dA, y = rn(B, H, W, 64).bfloat16(), rn(B, H, W, 64).bfloat16(), rn(B, H, W, 64).bfloat16()
    sc, sh = torch.rand(B, 64, device=dev) + 0.5, rn(B, 64) * 0.1
    gamma, nsc, nsh = torch.rand(64, device=dev) + 0.5, torch.rand(B, 64, device=dev) + 0.5, rn(B, 64) * 0.3
    rstd, mean, k1, k2 = torch.rand(B, 64, device=dev) + 0.5, rn(B, 64) * 0.2, rn(B, 64) * 0.1, rn(B, 64) * 0.1
    grad = torch.zeros(64, 64, 3, 3, device=dev)
    ws = torch.empty(L.lib().p4c_conv_wgrad_workspace_bytes(64, 3) // 4, dtype=torch.float32, device=dev)
    st = L.stream(dev)
    def call(transform, nb):
        a = (L.ptr(sc), L.ptr(sh), 1) if transform else (None, None, 0)
        if nb:
            L.call("p4c_conv_wgrad_nb", L.ptr(x), a[0], a[1], a[2], L.ptr(dA), L.ptr(y), L.ptr(gamma), L.ptr(nsc), L.ptr(nsh), L.ptr(rstd),
                   L.ptr(mean), L.ptr(k1), L.ptr(k2), 64, 64, L.ptr(grad), L.ptr(ws), B, H, W, st)
        else:
            L.call("p4c_conv_wgrad", L.ptr(x), L.BF16, L.BF16, 64, 3, a[0], a[1], a[2], L.ptr(dA), 64, 64, L.ptr(grad), L.ptr(ws), B, H, W, st)
    def run(n, transform, nb):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5): call(transform, nb)
        a.record()
        for _ in range(n): call(transform, nb)
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1000
    for rep in range(2):
        for name, env in (("tile kernel", {"P4C_NO_WGRAD_ROWS": "1"}), ("row kernel", {}), ("row kernel, 12 segments", {"P4C_WGROWS_NSEG": "12"}),
                          ("row kernel, 8 segments", {"P4C_WGROWS_NSEG": "8"})):
            for k in ("P4C_NO_WGRAD_ROWS", "P4C_WGROWS_NSEG"): os.environ.pop(k, None)
            os.environ.update(env)
            print(f"{B}x{H}x{W} {name:26s} launch + reduce [us]: plain {run(40, False, False):6.1f}  transform {run(40, True, False):6.1f}  "
                  f"NormBwdCoef {run(40, False, True):6.1f}  transform + NormBwdCoef {run(40, True, True):6.1f}", flush=True)
bench(2, 512, 512)
bench(2, 512, 640)
bench(2, 256, 256)
