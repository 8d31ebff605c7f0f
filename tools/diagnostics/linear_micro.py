"""Per-shape timing of the three GEMMs of a Linear at SwinUNetR's sizes: library (as selected in this process) vs the tall-skinny kernels."""
import sys, torch
sys.path.insert(0, ".")
import py4cast_amd
from py4cast_amd import _lib as L, ops_ts as TS
dev = "cuda"
L.require_cuda(torch.zeros(1, device=dev))
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"{'R':>7s} {'K':>4s} {'O':>4s} | fwd lib | dx lib | dW lib | fwd nat | dx nat | dW+db nat | MB(fwd)")
for R, K, O in ((131072, 24, 72), (131072, 24, 24), (131072, 24, 96), (131072, 96, 24), (32768, 48, 144), (32768, 48, 48), (32768, 48, 192), (32768, 192, 48),
                (8192, 96, 288), (8192, 96, 96), (8192, 96, 384), (8192, 384, 96), (2048, 192, 576), (524288, 24, 24), (32768, 128, 512), (32768, 512, 128), (32768, 128, 64), (32768, 64, 128), (8192, 256, 128), (8192, 128, 256), (8192, 256, 256)):
    x = torch.randn(R, K, device=dev, dtype=torch.bfloat16); dy = torch.randn(R, O, device=dev, dtype=torch.bfloat16)
    w = torch.randn(O, K, device=dev, dtype=torch.bfloat16); b = torch.randn(O, device=dev, dtype=torch.bfloat16)
    f = t(lambda: torch.nn.functional.linear(x, w, b))
    dx = t(lambda: dy @ w)
    dw = t(lambda: dy.t() @ x)
    from py4cast_amd import ops_rows as OR
    wf, bf = w.float(), b.float()
    if L.lib().p4c_row_gemm_supported(K, O) and L.lib().p4c_row_gemm_supported(O, K) and OR._wgrad_chunks(O, K, True) is not None:
        g = t(lambda: OR._row_gemm(x, wf, False, bf, O))
        a = t(lambda: OR._row_gemm(dy, wf, True, None, K))
        lib = L.lib()
        kc = min(K, 192)
        kp = 32 * ((kc + 1 + 31) // 32)
        out = torch.empty(64 * ((O + 63) // 64), kp, dtype=torch.float32, device=dev)
        ws = torch.empty(max(lib.p4c_row_gemm_wgrad_workspace_bytes(R, O, kc, 1) // 4, 1), dtype=torch.float32, device=dev)
        c = t(lambda: L.call("p4c_row_gemm_wgrad", L.ptr(dy), O, L.ptr(x), K, L.ptr(out), L.ptr(ws), R, O, kc, 1, L.stream(x.device)))   # (first <= 192 input features)
    else:
        g = a = c = float("nan")
    print(f"{R:7d} {K:4d} {O:4d} | {f:7.1f} | {dx:6.1f} | {dw:6.1f} | {g:7.1f} | {a:6.1f} | {c:9.1f} | {R * (K + O) * 2 / 1e6:6.1f}")
