import sys, torch, ctypes
sys.path.insert(0, '/root/repo')
from py4cast_amd import ops_model as om, _lib as L
dev = torch.device('cuda:0')
B, H, W = 2, 512, 512
x = torch.randn(B, H, W, 64, device=dev).bfloat16()
dout = torch.randn(B, H, W, 64, device=dev).bfloat16()
grad = torch.zeros(64, 64, 3, 3, device=dev)
buf = torch.zeros(4096, dtype=torch.int64, device=dev)
h = ctypes.CDLL(L.LIB_PATH)
for _ in range(50): om.conv_wgrad(x, dout, 3, 64, 64, grad, None, None, False, compute="bf16")
h.p4c_debug_set_stamps(ctypes.c_void_p(buf.data_ptr()))
om.conv_wgrad(x, dout, 3, 64, 64, grad, None, None, False, compute="bf16"); torch.cuda.synchronize()
h.p4c_debug_set_stamps(None)
s = buf.cpu().tolist()
cyc, rt = s[3002] - s[3000], s[3003] - s[3001]
print("loop: %d shader cycles, %d x10ns -> %.2f us, clock %.2f GHz" % (cyc, rt, rt / 100, cyc / (rt * 10) ))
for i in range(8):
    t = [s[2000 + 4 * i + k] - s[3000] for k in range(3)]
    print(" tile %d: start %6d  sync+store+loadissue %5d  mfma %5d" % (i, t[0], t[1] - t[0], t[2] - t[1]))
