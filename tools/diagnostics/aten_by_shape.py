import sys, collections, torch
sys.path.insert(0, ".")
sys.argv = ["bench.py", "--no-cpu-baseline"]
import bench as Bn
from py4cast_amd.lightning import AutoRegressiveLightning
from py4cast_amd.trainer import FlatDDP
from torch.profiler import ProfilerActivity, profile
device = torch.device("cuda", 0)
model, strategy, T = "UNetRPP", "diff_ar", 2
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
# where do the copies / casts of the full-resolution 64-channel maps come from?  (first frame of this package on the Python stack)
want = {"aten::copy_", "aten::_to_copy", "aten::contiguous", "aten::clone"}
by = collections.Counter()
for e in prof.events():
    if e.name in want and e.input_shapes and list(e.input_shapes[0]) in ([2, 64, 512, 512], [2, 512, 512, 64]):
        fr = next((f for f in (e.stack or []) if "py4cast_amd/" in f), "<no python frame: autograd>")
        by[(e.name, fr.split("py4cast_amd/")[-1][:70])] += 1
for (n, fr), c in by.most_common(14):
    print(f"{c:4d}x {n:16s} {fr}")
