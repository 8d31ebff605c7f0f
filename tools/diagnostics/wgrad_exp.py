import sys, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py4cast_amd import ops_model as om
dev = torch.device('cuda:0')
B, H, W = 2, 512, 512
x = torch.randn(B, H, W, 64, device=dev).bfloat16()
dout = torch.randn(B, H, W, 64, device=dev).bfloat16()
sc = torch.rand(B, 64, device=dev) + 0.5; sh = torch.randn(B, 64, device=dev) * 0.1
grad = torch.zeros(64, 64, 3, 3, device=dev)
def run(n, transform):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    args = (sc, sh, True) if transform else (None, None, False)
    for _ in range(5): om.conv_wgrad(x, dout, 3, 64, 64, grad, *args, compute="bf16")
    a.record()
    for _ in range(n): om.conv_wgrad(x, dout, 3, 64, 64, grad, *args, compute="bf16")
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1000
print("LIB", os.path.basename(os.environ.get("P4C_LIB_PATH", "default")), "wgrad+reduce plain %.1f us" % run(30, False), " transform %.1f us" % run(30, True))
