import sys, torch
sys.path.insert(0, '/root/repo')
import bench
from py4cast_amd.lightning import AutoRegressiveLightning
from py4cast_amd.trainer import FlatDDP, GraphedTrainingStep
dev = torch.device('cuda:0')
H = W = int(sys.argv[1]); model = sys.argv[2]
case = bench.synthetic_case(1234, 2, 3, 1, H, W, 60, 5, 4, 0, dev)
info = bench.make_info(case, 5)
torch.manual_seed(1234)
lm = AutoRegressiveLightning({"activation_dtype": "bf16"}, info, None, num_pred_steps_train=3, batch_size=2, model_name=model,
                             losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
                             training_strategy="scaled_ar").to(dev)
ddp = FlatDDP(lm.model, 1)
loss = lm.training_step(bench.make_batch(case), 0); loss.backward(); print("eager loss", float(loss)); del loss
ddp.zero_grad()
g = GraphedTrainingStep(lm, bench.make_batch(case))
ddp.zero_grad()
l = g(bench.make_batch(case)); torch.cuda.synchronize(); print("graph loss", float(l))
bad = [n for n, p in lm.model.named_parameters() if not torch.isfinite(p.grad).all()]
print("non-finite grads:", len(bad), bad[:12])
eager = {}
ddp.zero_grad()
loss = lm.training_step(bench.make_batch(case), 0); loss.backward(); del loss
torch.cuda.synchronize()
eager = {n: p.grad.clone() for n, p in lm.model.named_parameters()}
for rep in range(3):
    ddp.zero_grad()
    l = g(bench.make_batch(case)); torch.cuda.synchronize()
    worst = sorted(((float((p.grad - eager[n]).norm() / eager[n].norm().clamp_min(1e-30)), n, float(p.grad.abs().max())) for n, p in lm.model.named_parameters()), reverse=True)[:8]
    print("replay", rep, [(round(a, 4), n, m) for a, n, m in worst])
