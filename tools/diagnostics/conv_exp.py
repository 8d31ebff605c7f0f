import sys, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py4cast_amd import ops_model as om
dev = torch.device('cuda:0')
B, H, W = 2, 512, 512
x = torch.randn(B, H, W, 64, device=dev).bfloat16()
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
sc = torch.rand(B, 64, device=dev) + 0.5; sh = torch.randn(B, 64, device=dev) * 0.1
wp = om.prep_weights(w, False, 64, 64, compute="bf16")
def run(n, **kw):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): om.conv_fwd(x, wp, 3, compute="bf16", **kw)
    a.record()
    for _ in range(n): om.conv_fwd(x, wp, 3, compute="bf16", **kw)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1000
print("LIB", os.path.basename(os.environ.get("P4C_LIB_PATH", "default")), "plain %.1f us" % run(30), " transform+stats %.1f us" % run(30, in_scale=sc, in_shift=sh, in_relu=True, want_stats=True),
      " stats only %.1f us" % run(30, want_stats=True))
