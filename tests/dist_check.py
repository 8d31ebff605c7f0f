"""Child job of tests/test_dist_gpu.py (run under torch.distributed.run, one rank per GPU, or alone): three optimizer steps of the
native HalfUNet rollout on a fixed global batch of 4 samples, each rank taking its share; writes the final parameters and the
per-step global mean loss of rank 0's view to <out>/rank<r>.pt."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(out_dir, sharded):
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    # P4C_DIST_SHARE_GPU=1: every rank on cuda:0 with the gloo backend (device tensors staged through the host) -- what a 1-GPU box
    # can run: the ranks' kernels, FlatDDP's communication-stream logic and the sharded optimizer are the real ones, only the
    # transport is not RCCL
    share = os.environ.get("P4C_DIST_SHARE_GPU") == "1"
    local = 0 if share else local
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    from helpers import make_batch, make_dataset_info, synthetic_case
    from py4cast_amd.lightning import AutoRegressiveLightning
    from py4cast_amd.trainer import FlatDDP

    case = synthetic_case(seed=77, B=4, T=2, H=32, W=32, F=12, Ff=5, Fs=4, border=2)
    info = make_dataset_info(case, 5)
    per = 4 // world
    mine = {k: (v[rank * per:(rank + 1) * per] if k in ("inputs", "forcing", "outputs") else v) for k, v in case.items()}
    torch.manual_seed(100 + rank)          # different initial weights on purpose: FlatDDP broadcasts rank 0's
    lm = AutoRegressiveLightning({"norm": "group"}, info, None, num_pred_steps_train=2, batch_size=per, model_name="HalfUNet",
                                 losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
                                 training_strategy="scaled_ar", learning_rate=1e-3, num_warmup_steps=0).to(device).train()
    if world == 1:
        torch.manual_seed(100)             # the single-process run starts from rank 0's weights too
        ref = AutoRegressiveLightning({"norm": "group"}, info, None, num_pred_steps_train=2, batch_size=per, model_name="HalfUNet",
                                      training_strategy="scaled_ar")
        lm.model.load_state_dict(ref.model.state_dict())
    ddp = FlatDDP(lm.model, world, sharded=sharded)
    opt = lm.configure_optimizers()["optimizer"]
    losses = []
    for i in range(3):
        loss = lm.training_step(make_batch(mine, device), i)
        loss.backward()
        ddp.all_reduce_grads()
        if ddp.sharded:
            opt.step_shards(ddp.shards())
            ddp.all_gather_params()
        else:
            opt.step()
        ddp.zero_grad()
        l = loss.detach().clone()
        if world > 1:
            torch.distributed.all_reduce(l)
            l /= world
        losses.append(float(l))
    torch.save({"params": torch.cat([p.detach().reshape(-1) for p in lm.model.parameters()]).cpu(), "losses": losses},
               os.path.join(out_dir, f"rank{rank}_w{world}.pt"))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], len(sys.argv) > 2 and sys.argv[2] == "sharded")
