import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from py4cast_amd.halfunet import HalfUNetMI355X, HalfUNetSettings
dev = torch.device("cuda:0")
B, H, W = 2, 512, 512
x = torch.randn(B, H, W, 69, device=dev)
def run(tag):
    torch.manual_seed(3)
    m = HalfUNetMI355X(69, 60, (H, W), HalfUNetSettings(norm="batch", compute_dtype="bf16", activation_dtype="bf16")).to(dev).train()
    with torch.no_grad():
        for _ in range(3): m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): m(x)
        e1.record(); torch.cuda.synchronize()
    print(tag, round(e0.elapsed_time(e1) / 20 * 1e3, 1), "us per forward")
os.environ["P4C_COARSE_FWD"] = "0"; run("per-launch plan      ")
os.environ["P4C_COARSE_FWD"] = "1"
for exp, tag in ((0, "coarse kernel        "), (1, "  no conv tiles      "), (2, "  no pools           "), (3, "  no conv, no pools  "), (4, "  no fences          "), (7, "  barriers only      ")):
    os.environ["P4C_CF_EXP"] = str(exp); run(tag)
