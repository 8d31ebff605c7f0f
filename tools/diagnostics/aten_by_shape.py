import sys, collections, torch
sys.path.insert(0, ".")
sys.argv = ["bench.py", "--no-cpu-baseline"]
import bench as Bn
from py4cast_amd.lightning import AutoRegressiveLightning
from py4cast_amd.trainer import FlatDDP
from torch.profiler import ProfilerActivity, profile
device = torch.device("cuda", 0)
import os
model = os.environ.get("MODEL", "UNetRPP")
strategy, T = ("diff_ar", 2) if model == "UNetRPP" else ("scaled_ar", 3)
B, F, Ff, Fs, H, W = 2, 60, 5, 4, 512, 512
case = Bn.synthetic_case(1234, B, T, 1, H, W, F, Ff, Fs, 0, device)
info = Bn.make_info(case, Ff)
torch.manual_seed(1234)
lm = AutoRegressiveLightning(Bn.model_settings(model, "bf16"), info, None, num_input_steps=1, num_pred_steps_train=T, num_pred_steps_val_test=T,
                             batch_size=B, model_name=model, losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
                             training_strategy=strategy, learning_rate=1e-3).to(device)
ddp = FlatDDP(lm.model, 1)
for _ in range(2):
    ddp.zero_grad(); lm.training_step(Bn.make_batch(case), 0).backward()
torch.cuda.synchronize(); ddp.zero_grad()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    lm.training_step(Bn.make_batch(case), 0).backward(); torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    t = getattr(ev, "device_time_total", 0)
    if ev.key in ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::stack", "aten::_softmax", "aten::_softmax_backward_data", "aten::mul", "aten::sum", "aten::fill_", "aten::zero_", "aten::_to_copy") and t > 0:
        rows.append((t, ev.count, ev.key, str(ev.input_shapes)[:80]))
for t, n, k, shp in sorted(rows, reverse=True)[:26]:
    print(f"{t/1e3:7.2f} ms {n:4d}x {k:28s} {shp}")
# where do the copies / casts come from?  (first frame of this package on the Python stack, per input shape)
want = {"aten::copy_", "aten::_to_copy", "aten::contiguous", "aten::clone", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::cat", "aten::mul"}
by = collections.Counter()
tm = collections.Counter()
for e in prof.events():
    if e.name in want and e.input_shapes and getattr(e, "device_time_total", 0) > 0:
        fr = next((f for f in (e.stack or []) if "py4cast_amd/" in f), "<no python frame: autograd>")
        k = (e.name, str(list(e.input_shapes[0]))[:28], fr.split("py4cast_amd/")[-1][:60])
        by[k] += 1
        tm[k] += e.device_time_total
for k, t in tm.most_common(40):
    print(f"{t/1e3:7.3f} ms {by[k]:4d}x {k[0]:16s} {k[1]:28s} {k[2]}")
