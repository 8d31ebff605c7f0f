"""(Historical: written against a build whose EPA added the token-projection bias with a broadcasting `+`, switched on by
P4C_TMP_PLUSBIAS=1 -- that switch is gone; the pure-PyTorch reproducers replay_reduce_micro*.py and memset_node_probe.py supersede it.)
UNetRPP at the bench sizes: when does the replayed gradient of E.bias go wrong -- after a parameter change, or after any eager pass?"""
import os, sys, torch
os.environ["P4C_TMP_PLUSBIAS"] = "1"
sys.path.insert(0, ".")
import bench as Bn
from py4cast_amd import _lib as L
from py4cast_amd.lightning import AutoRegressiveLightning
from py4cast_amd.trainer import FlatDDP, GraphedTrainingStep
device = torch.device("cuda", 0)
B, F, T, Ff, Fs, H, W = 2, 60, 6, 5, 4, 512, 512
case = Bn.synthetic_case(1234, B, T, 1, H, W, F, Ff, Fs, 0, device)
info = Bn.make_info(case, Ff)
settings = {"hidden_size": 1024, "num_heads_encoder": 16, "num_heads_decoder": 4, "depths": [3, 3, 3, 3], "linear_upsampling": True, "downsampling_rate": 4,
            "decoder_proj_size": 64, "encoder_proj_sizes": [64, 64, 64, 32], "attention_code": "torch", "activation_dtype": "bf16"}
torch.manual_seed(1234)
lm = AutoRegressiveLightning(settings, info, None, num_input_steps=1, num_pred_steps_train=T, num_pred_steps_val_test=T, batch_size=B, model_name="UNetRPP",
                             losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
                             training_strategy="diff_ar", learning_rate=1e-3).to(device)
ddp = FlatDDP(lm.model, 1)
ddp.zero_grad()
batch = Bn.make_batch(case)
step = GraphedTrainingStep(lm, batch, verify=False)
name = "decoder5.decoder_block.0.0.epa_block.E.bias"
p = dict(lm.model.named_parameters())[name]
w = dict(lm.model.named_parameters())["decoder5.decoder_block.0.0.epa_block.E.weight"]
def replay():
    ddp.zero_grad(); step(batch); torch.cuda.synchronize(); return p.grad.detach().clone(), w.grad.detach().clone()
def eager():
    ddp.zero_grad(); lm.training_step(Bn.make_batch(case), 0).backward(); torch.cuda.synchronize(); return p.grad.detach().clone(), w.grad.detach().clone()
rel = lambda a, b: float((a - b).norm() / b.norm())
r0 = replay(); r1 = replay()
print("replay vs replay (nothing in between):            bias %.3g  weight %.3g" % (rel(r1[0], r0[0]), rel(r1[1], r0[1])))
e0 = eager()
print("eager vs replay:                                  bias %.3g  weight %.3g" % (rel(e0[0], r0[0]), rel(e0[1], r0[1])))
r2 = replay()
print("replay after an eager pass (same parameters):     bias %.3g  weight %.3g" % (rel(r2[0], r0[0]), rel(r2[1], r0[1])))
with torch.no_grad():
    torch._foreach_mul_(list(lm.model.parameters()), 1.0 + 2.0 ** -7)
L.PARAM_EPOCH[0] += 1
r3 = replay()
e3 = eager()
print("after a parameter update: replay vs eager:        bias %.3g  weight %.3g   |bias grad| replay %.3g eager %.3g" % (rel(r3[0], e3[0]), rel(r3[1], e3[1]), float(r3[0].norm()), float(e3[0].norm())))
r4 = replay()
print("second replay after the update vs eager:          bias %.3g  weight %.3g" % (rel(r4[0], e3[0]), rel(r4[1], e3[1])))
only = [q for n, q in lm.model.named_parameters() if n != name]
with torch.no_grad():
    torch._foreach_mul_(only, 1.0 / (1.0 + 2.0 ** -7))     # everything back except this bias
r5 = replay(); e5 = eager()
print("all parameters back except this bias:             bias %.3g  weight %.3g" % (rel(r5[0], e5[0]), rel(r5[1], e5[1])))
