"""Which SOURCE LINES of the package issue a model's torch (non-native) kernels?  torch.profiler with stacks over one eager training
step; device time of every ATen op grouped by the innermost py4cast_amd frame of its Python stack (backward ops that autograd runs
without a Python frame are listed under "<autograd>" with their op name and input shapes).
usage: op_sources.py [MODEL] [strategy] [pred_steps]"""
import collections
import sys

import torch

sys.path.insert(0, ".")
import bench as Bn
from py4cast_amd.lightning import AutoRegressiveLightning
from py4cast_amd.trainer import FlatDDP
from torch.profiler import ProfilerActivity, profile

device = torch.device("cuda", 0)
model = sys.argv[1] if len(sys.argv) > 1 else "SwinUNetR"
strategy = sys.argv[2] if len(sys.argv) > 2 else "scaled_ar"
T = int(sys.argv[3]) if len(sys.argv) > 3 else 3
B, F, Ff, Fs, H, W = 2, 60, 5, 4, 512, 512
case = Bn.synthetic_case(1234, B, T, 1, H, W, F, Ff, Fs, 0, device)
info = Bn.make_info(case, Ff)
torch.manual_seed(1234)
lm = AutoRegressiveLightning(Bn.model_settings(model, "bf16"), info, None, num_input_steps=1, num_pred_steps_train=T, num_pred_steps_val_test=T,
                             batch_size=B, model_name=model,
                             losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
                             training_strategy=strategy, learning_rate=1e-3).to(device)
ddp = FlatDDP(lm.model, 1)
for _ in range(2):
    ddp.zero_grad()
    lm.training_step(Bn.make_batch(case), 0).backward()
torch.cuda.synchronize()
ddp.zero_grad()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    lm.training_step(Bn.make_batch(case), 0).backward()
    torch.cuda.synchronize()
by_line = collections.defaultdict(lambda: [0.0, 0, collections.Counter()])
total = native = 0.0
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
    t = getattr(ev, "self_device_time_total", 0)
    if t <= 0:
        continue
    total += t
    if "p4c" in ev.key or ev.key.startswith("_") and not ev.key.startswith("__amd"):
        native += t
        continue
    where = "<autograd> " + ev.key + " " + str(ev.input_shapes)[:70]
    for fr in ev.stack or []:
        if "py4cast_amd/" in fr:
            where = fr.split("py4cast_amd/")[1].strip()
            break
    e = by_line[where]
    e[0] += t
    e[1] += ev.count
    e[2][ev.key] += t
print(f"{model}: self device time of one training step {total / 1e3:.1f} ms, native kernels / nodes {native / 1e3:.1f} ms")
for where, (t, n, ops) in sorted(by_line.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{t / 1e3:7.2f} ms {n:5d}x  {where[:100]:100s} {', '.join(f'{k}:{v / 1e3:.2f}' for k, v in ops.most_common(3))}")
