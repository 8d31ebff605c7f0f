"""Per-launch time of the row-streaming 3x3 conv for ONE library build (P4C_LIB_PATH), both MFMA shapes: used for the
stage-removal runs (tools/diagnostics/rows_build.sh expN -DP4C_EXP=N)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py4cast_amd import ops_model as om

dev = torch.device("cuda:0")
B, H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2x512x512").split("x"))
x = torch.randn(B, H, W, 64, device=dev).bfloat16()
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
sc = torch.rand(B, 64, device=dev) + 0.5
sh = torch.randn(B, 64, device=dev) * 0.1
wp = om.prep_weights(w, False, 64, 64, compute="bf16")


def run(n, **kw):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        om.conv_fwd(x, wp, 3, compute="bf16", **kw)
    a.record()
    for _ in range(n):
        om.conv_fwd(x, wp, 3, compute="bf16", **kw)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1000


cases = {"plain": {}, "stats": dict(want_stats=True), "transform+stats": dict(in_scale=sc, in_shift=sh, in_relu=True, want_stats=True)}
out = []
for mf in ("16", "32"):
    os.environ["P4C_ROWS_MFMA"] = mf
    out.append("mfma%s: " % mf + "  ".join("%s %.1f" % (n, min(run(30, **kw) for _ in range(3))) for n, kw in cases.items()))
print("%-22s %dx%dx%d  %s" % (os.path.basename(os.environ.get("P4C_LIB_PATH", "default")), B, H, W, " | ".join(out)))
