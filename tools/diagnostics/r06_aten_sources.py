"""Where do the library (non-p4c) kernels of a training step come from?  One EAGER step of a bench.py configuration under
torch.profiler (CPU + device activities, Python stacks): every device kernel is attributed to the innermost CPU op that launched it, that
op to its enclosing ops and to the innermost py4cast_amd source line -- the list VERDICT r5 item 4 asks to empty (UNETR++ / Swin glue).
usage: python tools/diagnostics/r06_aten_sources.py <model> [pred steps] [strategy] [top n]"""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "UNetRPP"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 6
strategy = sys.argv[3] if len(sys.argv) > 3 else "diff_ar"
top = int(sys.argv[4]) if len(sys.argv) > 4 else 45
dev = torch.device("cuda:0")
from py4cast_amd.lightning import AutoRegressiveLightning  # noqa: E402
from py4cast_amd.trainer import FlatDDP  # noqa: E402

case = bench.synthetic_case(1234, 2, T, 1, 512, 512, 60, 5, 4, 0, dev)
info = bench.make_info(case, 5)
torch.manual_seed(1234)
lm = AutoRegressiveLightning(bench.model_settings(model, "bf16"), info, None, num_input_steps=1, num_pred_steps_train=T,
                             num_pred_steps_val_test=T, batch_size=2, model_name=model,
                             losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
                             training_strategy=strategy, learning_rate=1e-3, min_learning_rate=3e-7, num_warmup_steps=1000,
                             betas=(0.9, 0.95)).to(dev)
ddp = FlatDDP(lm.model, 1)
lm.use_param_proxies = True          # what the captured step runs with (per-AR-step stand-ins of the parameters)


def step(i):
    loss = lm.training_step(bench.make_batch(case), i)
    loss.backward()
    ddp.zero_grad()


for i in range(2):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(2)
    torch.cuda.synchronize()

rows = defaultdict(lambda: [0.0, 0])
total = native = 0.0
for ev in prof.events():
    if "DeviceType.CPU" not in str(ev.device_type) or not ev.kernels:
        continue
    if any(c.kernels for c in ev.cpu_children):
        continue
    frame = next((f for f in (ev.stack or []) if "py4cast_amd" in f), "")
    frame = frame.split("py4cast_amd/")[-1][:70]
    chain, p = [], ev.cpu_parent
    while p is not None and len(chain) < 5:
        chain.append(p.name[:40].replace("autograd::engine::evaluate_function: ", "eval:"))
        p = p.cpu_parent
    for k in ev.kernels:
        total += k.duration
        if "p4c" in k.name:
            native += k.duration
            continue
        short = k.name.replace("void at::native::", "").replace("(anonymous namespace)::", "")
        for w in ("direct_copy_kernel", "CUDAFunctor_add", "FillFunctor", "MulFunctor", "sum_functor", "bfloat16tofloat32", "bfloat16_copy",
                  "CatArray", "multi_tensor_apply", "bernoulli", "Cijk", "copyBuffer"):
            if w in short:
                short = w + (" bf16" if "BFloat16" in k.name else " f32" if "float" in k.name else "")
                break
        shp = str([tuple(x) for x in (ev.input_shapes or []) if x])[:60]
        key = (short[:40], ev.name[:40], " < ".join(chain), frame or shp)
        rows[key][0] += k.duration
        rows[key][1] += 1
print(f"{model} T={T} {strategy}: device kernel time {total / 1e3:.2f} ms, native {native / total:.3f}")
for key, (t, n) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{t / 1e3:8.3f} ms {n:5d}  {key[0]:40s} | {key[1]:40s} | {key[2][:150]:150s} | {key[3]}")
