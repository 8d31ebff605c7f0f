"""Round 4 robustness sweep (GPU): a 2-step rollout training step of the bf16 HalfUNet through the native rollout at odd shapes / feature
counts, every round-4 fusion ON (defaults) against every one OFF (P4C_FUSED_TAIL=0, P4C_FUSED_OUT_BWD=0, P4C_NO_FLAT_STEP=1,
P4C_BWD_INFIN_MAX=0): finite, same loss (the forward's bits do not depend on the fusions), gradient cosine."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from py4cast_amd.lightning import AutoRegressiveLightning

MSE = [{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}]
L1 = [{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "L1Loss", "reduction": "none"}}]
OFF = {"P4C_FUSED_TAIL": "0", "P4C_FUSED_OUT_BWD": "0", "P4C_NO_FLAT_STEP": "1", "P4C_BWD_INFIN_MAX": "0"}
dev = torch.device("cuda:0")
cases = [(1, 64, 64, 60, 5), (3, 80, 112, 21, 21), (2, 48, 208, 7, 3), (4, 128, 128, 13, 1), (2, 96, 64, 62, 2), (1, 160, 176, 4, 4),
         (2, 64, 320, 33, 9), (5, 32, 48, 60, 5), (2, 512, 640, 60, 5)]
# (normalisation, loss, strategy) walked along with the shapes
flavours = [("batch", MSE, "scaled_ar"), ("group", MSE, "scaled_ar"), ("batch", L1, "diff_ar"), ("group", L1, "scaled_ar"), ("batch", MSE, "diff_ar")]
bad = 0
for ci, (B, H, W, F, Ff) in enumerate(cases):
    norm, losses, strategy = flavours[ci % len(flavours)]
    case = bench.synthetic_case(17, B, 2, 1, H, W, F, Ff, 4, 4, dev)
    info = bench.make_info(case, Ff)
    res = {}
    for mode in ("on", "off"):
        for k, v in OFF.items():
            if mode == "off":
                os.environ[k] = v
            else:
                os.environ.pop(k, None)
        torch.manual_seed(5)
        lm = AutoRegressiveLightning({"compute_dtype": "bf16", "activation_dtype": "bf16", "norm": norm}, info, None, num_pred_steps_train=2,
                                     batch_size=B, model_name="HalfUNet", losses=losses, training_strategy=strategy).to(dev).train()
        loss = lm.training_step(bench.make_batch(case), 0)
        loss.backward()
        torch.cuda.synchronize()
        g = torch.cat([q.grad.flatten() for q in lm.model.parameters()]).double()
        res[mode] = (float(loss.detach()), g)
    (l1, g1), (l0, g0) = res["on"], res["off"]
    cos = float(torch.dot(g1, g0) / (g1.norm() * g0.norm()))
    ok = torch.isfinite(g1).all().item() and abs(l1 - l0) <= 2e-6 * abs(l0) and cos > 0.98
    bad += not ok
    print(f"B={B} {H}x{W} F={F} Ff={Ff} {norm}norm {losses[0]['params']['loss']} {strategy}: loss {l1:.6f} / {l0:.6f}  grad cosine {cos:.5f}  {'ok' if ok else 'MISMATCH'}")
for k in OFF:
    os.environ.pop(k, None)
print("sweep:", "all ok" if not bad else f"{bad} mismatches")
sys.exit(1 if bad else 0)
