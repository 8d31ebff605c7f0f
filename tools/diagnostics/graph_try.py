import sys, time, torch
sys.path.insert(0, '/root/repo')
sys.argv = ['bench.py', '--no-cpu-baseline']
import bench
args = bench.parse_args()
from py4cast_amd.lightning import AutoRegressiveLightning
from py4cast_amd.trainer import FlatDDP
dev = torch.device('cuda:0')
H, W = args.grid; B, F, T, Ff, Fs = args.batch, args.features, args.pred_steps, 5, 4
case = bench.synthetic_case(1234, B, T, 1, H, W, F, Ff, Fs, args.border, dev)
info = bench.make_info(case, Ff)
lm = AutoRegressiveLightning({"compute_dtype": "bf16", "activation_dtype": "bf16"}, info, None, num_input_steps=1, num_pred_steps_train=T,
    num_pred_steps_val_test=T, batch_size=B, model_name="HalfUNet",
    losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
    training_strategy="scaled_ar", learning_rate=1e-3, min_learning_rate=3e-7, num_warmup_steps=1000, betas=(0.9, 0.95)).to(dev)
ddp = FlatDDP(lm.model, 1)
opt = lm.configure_optimizers()["optimizer"]
batch = bench.make_batch(case)
def fwd_bwd():
    loss = lm.training_step(batch, 0); loss.backward(); return loss
def step():
    loss = fwd_bwd(); opt.step(); ddp.zero_grad(); return loss
for i in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20): step()
torch.cuda.synchronize(); print("eager ms/step %.3f" % ((time.perf_counter() - t0) / 20 * 1e3))
# capture forward + backward (the optimizer's python-side step counter stays outside the graph)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(s):
        for i in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = fwd_bwd()
    def gstep():
        g.replay(); opt.step(); ddp.zero_grad()
    for i in range(3): gstep()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): gstep()
    torch.cuda.synchronize(); print("graph ms/step %.3f  loss %.4f" % ((time.perf_counter() - t0) / 20 * 1e3, float(static_loss)))
except Exception as e:
    print("graph capture failed:", type(e).__name__, str(e)[:300])
