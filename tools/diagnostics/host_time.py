import sys, time, torch, cProfile, pstats
sys.path.insert(0, "/root/repo" if __import__("os").path.exists("/root/repo/bench.py") else ".")
sys.argv = ["bench.py", "--no-cpu-baseline"]
import bench as Bn
import gc
from py4cast_amd.lightning import AutoRegressiveLightning
from py4cast_amd.trainer import FlatDDP
device = torch.device("cuda", 0)
B, F, T, Ff, Fs, H, W = 2, 60, 3, 5, 4, 512, 512
case = Bn.synthetic_case(1234, B, T, 1, H, W, F, Ff, Fs, 0, device)
info = Bn.make_info(case, Ff)
lm = AutoRegressiveLightning(Bn.model_settings("HalfUNet", "bf16"), info, None, num_input_steps=1, num_pred_steps_train=T, num_pred_steps_val_test=T, batch_size=B,
                             model_name="HalfUNet", losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
                             training_strategy="scaled_ar", learning_rate=1e-3).to(device)
ddp = FlatDDP(lm.model, 1)
opt = lm.configure_optimizers()
opt = opt["optimizer"] if isinstance(opt, dict) else (opt[0][0] if isinstance(opt, tuple) else opt)
def step():
    ddp.zero_grad()
    loss = lm.training_step(Bn.make_batch(case), 0)
    loss.backward()
    opt.step()
    return loss.detach()
for _ in range(10): step()
torch.cuda.synchronize(); gc.collect(); gc.disable()
for n in (3, 3, 6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} steps: host enqueue {1e3*(t1-t0)/n:.3f} ms per step, until done {1e3*(t2-t0)/n:.3f} ms per step")
# split: forward / backward / optimizer host time
tf = tb = to = 0.0
torch.cuda.synchronize()
for _ in range(4):
    torch.cuda.synchronize()
    a = time.perf_counter(); ddp.zero_grad(); loss = lm.training_step(Bn.make_batch(case), 0); b = time.perf_counter(); loss.backward(); c = time.perf_counter(); opt.step(); d = time.perf_counter()
    tf += b - a; tb += c - b; to += d - c
print(f"host: forward {1e3*tf/4:.3f} ms, backward {1e3*tb/4:.3f} ms, optimizer {1e3*to/4:.3f} ms (queue empty at the start of each step)")
pr = cProfile.Profile(); torch.cuda.synchronize(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
