"""Two eager training steps on one batch and one set of weights: where do they differ?  Loss, per-parameter gradients, and the
first module (forward order) whose output differs / the first (backward order) whose input gradient differs.
usage: [HOOKS=1] [DETERMINISTIC=1] determinism_probe.py MODEL [H W] [dtype]"""
import sys, os, torch
sys.path.insert(0, ".")
import bench as Bn
from py4cast_amd.lightning import AutoRegressiveLightning
from py4cast_amd.trainer import FlatDDP
model = sys.argv[1]
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 512)
dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
device = torch.device("cuda", 0)
if os.environ.get("DETERMINISTIC"):   # 1: library convolutions on their deterministic solvers; 2: also torch ops on their deterministic paths
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
    if os.environ["DETERMINISTIC"] == "2":
        torch.use_deterministic_algorithms(True, warn_only=True)
B, F, T, Ff, Fs = 2, 21, 3, 5, 4
case = Bn.synthetic_case(1234, B, T, 1, H, W, F, Ff, Fs, 0, device)
info = Bn.make_info(case, Ff)
settings = {"activation_dtype": dtype}
if model.lower().startswith("halfunet"):
    settings = {"compute_dtype": dtype, "activation_dtype": dtype}
if model.lower().startswith("unetrpp"):
    settings = {"hidden_size": 1024, "num_heads_encoder": 16, "num_heads_decoder": 4, "depths": [3, 3, 3, 3], "linear_upsampling": True,
                "downsampling_rate": 4, "decoder_proj_size": 64, "encoder_proj_sizes": [64, 64, 64, 32], "attention_code": "torch", "activation_dtype": dtype}
torch.manual_seed(1234)
lm = AutoRegressiveLightning(settings, info, None, num_input_steps=1, num_pred_steps_train=T, num_pred_steps_val_test=T, batch_size=B, model_name=model,
                             losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
                             training_strategy="diff_ar", learning_rate=1e-3).to(device)
ddp = FlatDDP(lm.model, 1)
fwd, bwd = [{}, {}], [{}, {}]
order_f, order_b = [], []
run = [0]
def fh(name):
    def f(m, i, o):
        if isinstance(o, torch.Tensor):
            k = name + "#" + str(sum(1 for n in fwd[run[0]] if n.startswith(name + "#")))
            fwd[run[0]][k] = o.detach().float().clone()
            if run[0] == 0:
                order_f.append(k)
    return f
def bh(name):
    def f(m, gi, go):
        if gi and isinstance(gi[0], torch.Tensor):
            k = name + "#" + str(sum(1 for n in bwd[run[0]] if n.startswith(name + "#")))
            bwd[run[0]][k] = gi[0].detach().float().clone()
            if run[0] == 0:
                order_b.append(k)
    return f
small = H * W <= 128 * 128
if small or os.environ.get("HOOKS"):
    for n, m in lm.model.named_modules():
        if n and len(list(m.children())) > 0 or n.count(".") <= 2:
            if n:
                m.register_forward_hook(fh(n)); m.register_full_backward_hook(bh(n))
grads, losses = [], []
for r in range(2):
    run[0] = r
    ddp.zero_grad()
    loss = lm.training_step(Bn.make_batch(case), 0)
    loss.backward()
    torch.cuda.synchronize()
    losses.append(float(loss.detach()))
    grads.append({n: p.grad.detach().float().clone() for n, p in lm.model.named_parameters() if p.grad is not None})
    del loss
print(model, (H, W), dtype, "losses", losses)
rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
errs = sorted(((rel(grads[1][n], grads[0][n]), n, float(grads[0][n].norm())) for n in grads[0]), reverse=True)
print("parameters:", len(errs), "  differing:", sum(e[0] > 0 for e in errs), "  > 1e-3:", sum(e[0] > 1e-3 for e in errs), "  > 5e-2:", sum(e[0] > 5e-2 for e in errs))
for e in errs[:8]:
    print("   ", e)
for k in order_f:
    if k in fwd[1] and not torch.equal(fwd[0][k], fwd[1][k]):
        print("first forward output that differs:", k, rel(fwd[1][k], fwd[0][k])); break
else:
    print("forward outputs of all hooked modules identical" if order_f else "(no hooks)")
for k in order_b:
    if k in bwd[1] and not torch.equal(bwd[0][k], bwd[1][k]):
        print("first input gradient (backward order) that differs:", k, rel(bwd[1][k], bwd[0][k])); break
else:
    print("input gradients of all hooked modules identical" if order_b else "")
