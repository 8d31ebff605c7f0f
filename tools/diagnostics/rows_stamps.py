"""In-kernel timeline of conv3x3_bf16_rows (build with tools/diagnostics/rows_build.sh stamps -DP4C_STAMPS, run with
P4C_LIB_PATH=tools/diagnostics/libs/lib_rows_stamps.so): per-row cycles of compute wave 0 and per-interval phases of loader
wave 0 of workgroup 7, the in-kernel clock (s_memtime / s_memrealtime)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py4cast_amd import ops_model as om, _lib as L

dev = torch.device("cuda:0")
B, H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2x512x512").split("x"))
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"
x = torch.randn(B, H, W, 64, device=dev).bfloat16()
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
sc = torch.rand(B, 64, device=dev) + 0.5
sh = torch.randn(B, 64, device=dev) * 0.1
wp = om.prep_weights(w, False, 64, 64, compute="bf16")
kw = {"plain": {}, "stats": dict(want_stats=True), "ts": dict(in_scale=sc, in_shift=sh, in_relu=True, want_stats=True)}[mode]
buf = torch.zeros(4096, dtype=torch.int64, device=dev)
h = ctypes.CDLL(L.LIB_PATH)
for _ in range(int(os.environ.get("WARM", "200"))):   # WARM=50000: ~2 s of back-to-back launches before the stamped ones (DVFS settled)
    om.conv_fwd(x, wp, 3, compute="bf16", **kw)
torch.cuda.synchronize()
h.p4c_debug_set_rows_stamps(ctypes.c_void_p(buf.data_ptr()))
for _ in range(20):
    om.conv_fwd(x, wp, 3, compute="bf16", **kw)
torch.cuda.synchronize()
h.p4c_debug_set_rows_stamps(None)
s = buf.cpu().tolist()
cyc = lambda i: s[2 * i]
rt = lambda i: s[2 * i + 1]
t0 = min(rt(0), rt(9))
us = lambda i: (rt(i) - t0) / 100.0
print("%s %dx%dx%d" % (mode, B, H, W))
print("loader: entry %.2f us, first group staged %.2f, last drain issued %.2f" % (us(0), us(1), us(2)))
print("compute: entry %.2f us, weights loaded / at barrier %.2f, released %.2f, end %.2f" % (us(9), us(10), us(11), us(12)))
nrows = 0
while cyc(300 + 2 * nrows) and nrows < 150:
    nrows += 1
k = 0
print("barriers: arrival (us) of compute waves 0..3, release - arrival (us) | loader wave 0 arrival, wait")
while cyc(700 + 8 * k) and k < 100:
    arr = [us(700 + 8 * k + 2 * w) for w in range(4)]
    rel = [us(701 + 8 * k + 2 * w) for w in range(4)]
    print("  barrier %2d: arrive %s  wait %s | loader arrives %.2f waits %.2f" % (
        k, " ".join("%.2f" % a for a in arr), " ".join("%.2f" % (r - a) for r, a in zip(rel, arr)), us(102 + 4 * k), us(103 + 4 * k) - us(102 + 4 * k)))
    k += 1
if nrows == 0:
    sys.exit(0)
clock = (cyc(301 + 2 * (nrows - 1)) - cyc(300)) / max(1, (rt(301 + 2 * (nrows - 1)) - rt(300))) * 100.0
print("rows %d, in-kernel clock %.0f MHz" % (nrows, clock))
for m in range(nrows):
    gap = (cyc(300 + 2 * (m + 1)) - cyc(301 + 2 * m)) if m + 1 < nrows else 0
    print("  row %3d: start %.2f us  body %5d cycles  gap to next %5d cycles" % (m, us(300 + 2 * m), cyc(301 + 2 * m) - cyc(300 + 2 * m), gap))
k = 0
while cyc(100 + 4 * k) and k < 200:
    a, b_, c, d = (cyc(100 + 4 * k + j) for j in range(4))
    print("  loader interval %2d: start %.2f us  ring store %5d  load+drain %5d  barrier wait %5d cycles" % (k, us(100 + 4 * k), b_ - a, c - b_, d - c))
    k += 1
