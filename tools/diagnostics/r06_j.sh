#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06j; mkdir -p $O
python3 - > $O/standins_debug.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from py4cast_amd.lightning import AutoRegressiveLightning
from tests.helpers import make_batch, make_dataset_info, synthetic_case
dev = torch.device("cuda:0")
H = W = 64; F, Ff, T = 6, 5, 3
case = synthetic_case(seed=77, B=2, T=T, H=H, W=W, F=F, Ff=Ff, border=0)
info = make_dataset_info(case, Ff)
mse = [{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}]
settings = dict(activation_dtype="bf16", hidden_size=128, num_heads_encoder=2, num_heads_decoder=2, depths=[1, 1, 1, 1], encoder_proj_sizes=[16, 16, 8, 4],
                decoder_proj_size=16, linear_upsampling=True, attention_code="torch", conv8_dropout=0.0)
torch.manual_seed(78)
lm = AutoRegressiveLightning(settings, info, None, num_input_steps=1, num_pred_steps_train=T, batch_size=2, model_name="UNetRPP", losses=mse, training_strategy="diff_ar").to(dev).train()
for use in (True, False):
    lm.use_param_proxies = use
    lm.zero_grad(set_to_none=(use is False))
    loss = lm.training_step(make_batch(case, dev), 0)
    loss.backward()
    print(use, float(loss), [n for n, p in lm.model.named_parameters() if p.grad is None][:20])
PY
cat $O/standins_debug.txt | tail -8
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/full_suite.txt; cat $O/full_suite.txt
