import sys, collections, torch
sys.path.insert(0, ".")
sys.argv = ["bench.py", "--no-cpu-baseline"]
import bench as Bn
import py4cast_amd.swinunetr as S
import py4cast_amd.ops_rows as R
import py4cast_amd.ops_model as OM
from py4cast_amd.lightning import AutoRegressiveLightning
device = torch.device("cuda", 0)
B, F, Ff, Fs, H, W, T = 2, 60, 5, 4, 512, 512, 1
case = Bn.synthetic_case(1234, B, T, 1, H, W, F, Ff, Fs, 0, device)
info = Bn.make_info(case, Ff)
lm = AutoRegressiveLightning(Bn.model_settings("SwinUNetR", "bf16"), info, None, num_input_steps=1, num_pred_steps_train=T, num_pred_steps_val_test=T,
                             batch_size=B, model_name="SwinUNetR", losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
                             training_strategy="scaled_ar").to(device)
seen = collections.Counter()
o_conv, o_lin, o_ln = S._conv_hw, R.linear_nd, S._layer_norm
def conv(m, x):
    route = "compact" if OM.conv_nhwc_supported(x, m.weight) and OM._compact_ok(x, m.weight) else ("nhwc-padded" if OM.conv_nhwc_supported(x, m.weight) else "LIBRARY")
    seen[("conv", route, tuple(m.weight.shape), tuple(x.shape))] += 1
    return o_conv(m, x)
def lin(x, w, b=None):
    seen[("linear", R._row_gemm_mode(x, w, b) or "LIBRARY", tuple(w.shape), tuple(x.shape))] += 1
    return o_lin(x, w, b)
def ln(m, x):
    C = x.shape[-1]
    seen[("layernorm", "native" if (C * x.element_size()) % 16 == 0 and C * x.element_size() <= 1024 else "LIBRARY", tuple(x.shape))] += 1
    return o_ln(m, x)
S._conv_hw, R.linear_nd, S._layer_norm = conv, lin, ln
lm.training_step(Bn.make_batch(case), 0)
for k, v in sorted(seen.items(), key=lambda kv: (kv[0][1] not in ("LIBRARY", "nhwc-padded"), -kv[1])):
    print(v, k)
