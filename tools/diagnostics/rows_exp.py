"""Per-launch time of the 3x3 64->64 bf16 convolution: row-streaming kernel (conv_rows.hip) against the tile-ring kernel
(P4C_NO_ROWS=1), interleaved in one process on one device; checks both against each other first."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py4cast_amd import ops_model as om

dev = torch.device("cuda:0")
shapes = [(2, 512, 512), (2, 256, 256), (2, 128, 128), (2, 64, 64), (8, 512, 512)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, H, W in shapes:
    x = torch.randn(B, H, W, 64, device=dev).bfloat16()
    w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
    sc = torch.rand(B, 64, device=dev) + 0.5
    sh = torch.randn(B, 64, device=dev) * 0.1
    wp = om.prep_weights(w, False, 64, 64, compute="bf16")

    def run(n, **kw):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            om.conv_fwd(x, wp, 3, compute="bf16", **kw)
        a.record()
        for _ in range(n):
            om.conv_fwd(x, wp, 3, compute="bf16", **kw)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1000

    cases = {"plain": {}, "stats": dict(want_stats=True), "transform+stats": dict(in_scale=sc, in_shift=sh, in_relu=True, want_stats=True)}
    outs = {}
    libs = {"rows16": ("0", "16"), "rows32": ("0", "32"), "ring": ("1", "16")}

    def select(lib):
        os.environ["P4C_NO_ROWS"], os.environ["P4C_ROWS_MFMA"] = libs[lib]

    for lib in libs:
        select(lib)
        for name, kw in cases.items():
            r = om.conv_fwd(x, wp, 3, compute="bf16", **kw)
            outs[(lib, name)] = r if isinstance(r, tuple) else (r, None)
    for name in cases:
        o1, s1 = outs[("ring", name)]
        for lib in ("rows16", "rows32"):
            o0, s0 = outs[(lib, name)]
            d = (o0.float() - o1.float()).abs().max().item()
            msg = "max |%s - ring| = %.3g" % (lib, d)
            if s0 is not None:
                t0 = s0.reshape(B, -1, 2, 64).sum(1)
                t1 = s1.reshape(B, -1, 2, 64).sum(1)
                msg += ", statistics rel %.3g" % ((t0 - t1).abs().max() / t1.abs().max()).item()
            print("%dx%dx%d %-16s %s" % (B, H, W, name, msg))
    res = {(l, n): [] for l in libs for n in cases}
    for rnd in range(5):
        for lib in libs:
            select(lib)
            for name, kw in cases.items():
                res[(lib, name)].append(run(20, **kw))
    for name in cases:
        print("%dx%dx%d %-16s " % (B, H, W, name) + "   ".join("%s %.1f us (min %.1f)" % (lib, sorted(res[(lib, name)])[2], min(res[(lib, name)])) for lib in libs))
os.environ.pop("P4C_NO_ROWS", None)
os.environ.pop("P4C_ROWS_MFMA", None)
