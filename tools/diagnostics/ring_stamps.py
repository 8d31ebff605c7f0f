import sys, torch, ctypes, time
sys.path.insert(0, '/root/repo')
from py4cast_amd import ops_model as om, _lib as L
dev = torch.device('cuda:0')
B, H, W = 2, 512, 512
x = torch.randn(B, H, W, 64, device=dev).bfloat16()
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
wp = om.prep_weights(w, False, 64, 64, compute="bf16")
buf = torch.zeros(4096, dtype=torch.int64, device=dev)
h = ctypes.CDLL(L.LIB_PATH)
for _ in range(50): om.conv_fwd(x, wp, 3, compute="bf16")
torch.cuda.synchronize()
h.p4c_debug_set_stamps(ctypes.c_void_p(buf.data_ptr()))
om.conv_fwd(x, wp, 3, compute="bf16"); torch.cuda.synchronize()
h.p4c_debug_set_stamps(None)
s = buf.cpu().tolist()
t0 = min(s[3100], s[3110])
f = lambda i: (s[i] - t0) / 100.0
print("loader: entry %.2f us, staged first tile %.2f, after first barrier %.2f, loop end %.2f, last drain done %.2f" % (f(3100), f(3101), f(3102), f(3103), f(3104)))
print("compute: entry %.2f us, weights+barrier %.2f, end %.2f" % (f(3110), f(3111), f(3112)))
for i in range(16):
    print("  tile %2d: start %.2f  mfma+epilogue %.2f  barrier wait %.2f" % (i, f(3120 + 2 * i), f(3121 + 2 * i) - f(3120 + 2 * i), (f(3122 + 2 * i) if i < 15 else f(3112)) - f(3121 + 2 * i)))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): om.conv_fwd(x, wp, 3, compute="bf16")
b.record(); torch.cuda.synchronize()
print("per launch (back to back): %.1f us" % (a.elapsed_time(b) / 50 * 1000))
