"""Which source lines issue SwinUNetR's small torch kernels (copies, adds, fills, casts)?  torch.profiler with stacks, one eager training step."""
import sys, collections, torch
sys.path.insert(0, ".")
import bench as Bn
from py4cast_amd.lightning import AutoRegressiveLightning
from py4cast_amd.trainer import FlatDDP
from torch.profiler import profile, ProfilerActivity
device = torch.device("cuda", 0)
B, F, T, Ff, Fs, H, W = 2, 60, 3, 5, 4, 512, 512
case = Bn.synthetic_case(1234, B, T, 1, H, W, F, Ff, Fs, 0, device)
info = Bn.make_info(case, Ff)
torch.manual_seed(1234)
lm = AutoRegressiveLightning({"activation_dtype": "bf16"}, info, None, num_input_steps=1, num_pred_steps_train=T, num_pred_steps_val_test=T, batch_size=B,
                             model_name=sys.argv[1] if len(sys.argv) > 1 else "SwinUNetR",
                             losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
                             training_strategy="scaled_ar", learning_rate=1e-3).to(device)
ddp = FlatDDP(lm.model, 1)
for _ in range(2):
    ddp.zero_grad(); lm.training_step(Bn.make_batch(case), 0).backward()
torch.cuda.synchronize()
ddp.zero_grad()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    lm.training_step(Bn.make_batch(case), 0).backward()
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    t = getattr(ev, "self_device_time_total", 0)
    if t > 0:
        rows.append((t, ev.count, ev.key, str(ev.input_shapes)[:110]))
tot = sum(r[0] for r in rows)
print("self device time, one training step: %.1f ms" % (tot / 1e3))
byname = collections.Counter()
for t, n, k, shp in rows:
    byname[k] += t
for k, t in byname.most_common(25):
    print(f"   {t/1e3:7.2f} ms  {k}")
print()
for t, n, k, shp in sorted(rows, reverse=True)[:60]:
    print(f"{t/1e3:7.2f} ms {n:5d}x  {k:38s} {shp}")
