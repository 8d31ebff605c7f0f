"""Which python lines issue a model's small torch kernels?  torch.profiler with stacks over one eager training step, ATen ops grouped
by the innermost py4cast_amd frame.  Usage: python tools/diagnostics/op_stacks.py UNetRPP|SwinUNetR [hidden]"""
import collections
import sys

import torch

sys.path.insert(0, ".")
import bench as Bn
from py4cast_amd.lightning import AutoRegressiveLightning
from py4cast_amd.trainer import FlatDDP
from torch.profiler import ProfilerActivity, profile

device = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "UNetRPP"
B, F, T, Ff, Fs, H, W = 2, 60, 2, 5, 4, 512, 512
case = Bn.synthetic_case(1234, B, T, 1, H, W, F, Ff, Fs, 0, device)
info = Bn.make_info(case, Ff)
torch.manual_seed(1234)
settings = {"activation_dtype": "bf16"}
if name == "UNetRPP":
    settings.update(hidden_size=int(sys.argv[2]) if len(sys.argv) > 2 else 1024, num_heads_encoder=16, num_heads_decoder=4, linear_upsampling=True,
                    downsampling_rate=4, decoder_proj_size=64, encoder_proj_sizes=[64, 64, 64, 32], depths=[3, 3, 3, 3])
lm = AutoRegressiveLightning(settings, info, None, num_input_steps=1, num_pred_steps_train=T, num_pred_steps_val_test=T, batch_size=B, model_name=name,
                             losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
                             training_strategy="scaled_ar", learning_rate=1e-3).to(device)
ddp = FlatDDP(lm.model, 1)
for _ in range(2):
    ddp.zero_grad(); lm.training_step(Bn.make_batch(case), 0).backward()
torch.cuda.synchronize()
ddp.zero_grad()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    lm.training_step(Bn.make_batch(case), 0).backward()
    torch.cuda.synchronize()
by = collections.defaultdict(lambda: [0.0, 0])
tot = 0.0
for ev in prof.events():
    t = getattr(ev, "self_device_time_total", 0)
    if t <= 0 or not ev.name.startswith("aten::"):
        continue
    # nearest enclosing op that is not an aten op: a custom autograd Function, an autograd node, a module call
    par, owner = ev.cpu_parent, None
    while par is not None:
        if not par.name.startswith("aten::"):
            owner = par.name
            break
        par = par.cpu_parent
    key = (ev.name, (owner or "(top level python)")[:110])
    by[key][0] += t
    by[key][1] += 1
    tot += t
print("aten self device time of one training step (T=%d): %.2f ms" % (T, tot / 1e3))
for (n, f), (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{t / 1e3:7.3f} ms {c:5d}x {n:22s} {f}")
