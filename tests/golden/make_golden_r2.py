#!/usr/bin/env python3
"""
Round-2 golden vectors -- runs ONLY in the build container (needs /root/reference and, for the scheduler fixture,
``transformers``).  Same recipe as make_golden.py (the unmodified reference files executed through stubs of their
non-hot-path imports); covers the reference behaviours the first set left out:

  r2_downscaling_only.npz   _common_step with training_strategy="downscaling_only" (lightning.py:541-558, 611-621, 725-766)
  r2_combined_loss.npz      CombinedLoss of two weighted members (losses.py:268-307) + training loss + BPTT gradients
  r2_mask_ratio.npz         mask_tensor (lightning.py:769-785) alone and inside a 3-step rollout (one draw per model call)
  r2_scheduler.npz          transformers.get_cosine_with_min_lr_schedule_with_warmup as configured at lightning.py:453-458

    python tests/golden/make_golden_r2.py

The fixtures are data only; the reference never travels to the GPU box.
"""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stubs + case builders of the first fixture set)

NamedTensor, ItemBatch = mg.NamedTensor, mg.ItemBatch
GDIMS = ["batch", "timestep", "lat", "lon", "features"]


def new_module(lightning, case, model, strategy, K=1, mask_on_nan=False, mask_ratio=0, feat=None):
    B = case["inputs"].shape[0]
    F = case["inputs"].shape[-1]
    feat = feat or [f"f{i}" for i in range(F)]
    lm = lightning.AutoRegressiveLightning.__new__(lightning.AutoRegressiveLightning)
    torch.nn.Module.__init__(lm)
    lm.model = model
    lm.training_strategy, lm.num_inter_steps = strategy, K
    lm.channels_last, lm.mask_ratio, lm.mask_on_nan = False, mask_ratio, mask_on_nan
    lm.diff_stats = mg.StatsLike({n: {"std": case["diff_std"][i], "mean": case["diff_mean"][i]} for i, n in enumerate(feat)})
    lm.stats = mg.StatsLike({n: {"std": case["std"][i]} for i, n in enumerate(feat)})
    bm, st = case["border_mask"].clone(), case["statics"].clone()
    lm.register_buffer("border_mask", bm)
    lm.register_buffer("interior_mask", 1.0 - bm)
    lm.register_buffer("grid_static_features", st.unsqueeze(0).expand(B, *st.shape).clone())
    return lm


def dataset_info(lm, case, feat):
    class DI:
        state_weights = {n: float(case["state_weight"][i]) for i, n in enumerate(feat)}
        stats = lm.stats
        diff_stats = lm.diff_stats

    return DI


def downscaling_only(losses, lightning):
    """Output features aro_p0..4 <-> forcing features arp_p0..4 (+2 forcings without a counterpart): the name rule of
    lightning.py:548-557 pairs `name.split("_")[1:]`."""
    F, Ff, Fs = 5, 7, 4
    case = mg.make_case(2000, border=2)
    feat = [f"aro_p{i}" for i in range(F)]
    fnames = ["arp_x0", "arp_p3", "arp_p0", "arp_x1", "arp_p1", "arp_p4", "arp_p2"]   # shuffled: the index list matters
    cin = Fs + Ff
    g = torch.Generator().manual_seed(91)
    w, b = torch.randn(F, cin, 3, 3, generator=g) * 0.15, torch.randn(F, generator=g) * 0.1
    lm = new_module(lightning, case, mg.TinyConv(cin, F, w, b), "downscaling_only", feat=feat)
    batch = ItemBatch(NamedTensor(case["inputs"].clone(), GDIMS, feat), NamedTensor(case["forcing"].clone(), GDIMS, fnames),
                      NamedTensor(case["outputs"].clone(), GDIMS, feat))
    pred, tgt = lm._common_step(batch, 0, "train")
    mask, tgt_m = lm.get_mask_on_nan(tgt)
    loss = losses.WeightedLoss("MSELoss", reduction="none")
    loss.prepare(lm, lm.interior_mask, dataset_info(lm, case, feat))
    val = loss(pred, tgt_m, mask)
    train_loss = torch.mean(val)
    train_loss.backward()
    np.savez_compressed(
        os.path.join(HERE, "r2_downscaling_only.npz"),
        **{f"in_{k}": v.numpy() for k, v in case.items()}, in_w=w.numpy(), in_b=b.numpy(),
        out_prediction=pred.tensor.detach().numpy(), out_loss_wmse=val.detach().numpy(), out_train_loss=train_loss.detach().numpy(),
        out_grad_w=lm.model.w.grad.numpy(), out_grad_b=lm.model.b.grad.numpy(),
        out_common_features_idx=np.array(lm.common_features_idx),
        meta=np.array(repr(dict(strategy="downscaling_only", K=1, T_in=1, border=2, nan=0, layout="grid", feat=feat, fnames=fnames))),
    )
    print("wrote r2_downscaling_only", pred.tensor.shape, lm.common_features_idx)


def combined_loss(losses, lightning):
    F, Ff, Fs = 5, 7, 4
    conf = [
        {"class": "WeightedLoss", "weight": 0.7, "params": {"loss": "MSELoss", "reduction": "none"}},
        {"class": "WeightedLoss", "weight": 0.3, "params": {"loss": "L1Loss", "reduction": "none"}},
    ]
    out = {}
    for tag, nan in (("nonan", False), ("nan", True)):
        case = mg.make_case(2100 + int(nan), border=2, nan=nan)
        feat = [f"f{i}" for i in range(F)]
        cin = F + Fs + Ff + int(nan)
        g = torch.Generator().manual_seed(92)
        w, b = torch.randn(F, cin, 3, 3, generator=g) * 0.15, torch.randn(F, generator=g) * 0.1
        lm = new_module(lightning, case, mg.TinyConv(cin, F, w, b), "scaled_ar", mask_on_nan=nan)
        batch = ItemBatch(NamedTensor(case["inputs"].clone(), GDIMS, feat),
                          NamedTensor(case["forcing"].clone(), GDIMS, [f"g{i}" for i in range(Ff)]),
                          NamedTensor(case["outputs"].clone(), GDIMS, feat))
        pred, tgt = lm._common_step(batch, 0, "train")
        mask, tgt_m = lm.get_mask_on_nan(tgt)
        cl = losses.CombinedLoss(conf)
        cl.prepare(lm, lm.interior_mask, dataset_info(lm, case, feat))
        val = cl(pred, tgt_m, mask=mask)
        vmap = cl(pred, tgt_m, mask=mask, reduce_spatial_dim=False)
        train_loss = torch.mean(val)
        train_loss.backward()
        out.update({f"in_{tag}_{k}": v.numpy() for k, v in case.items()})
        out.update({f"in_{tag}_w": w.numpy(), f"in_{tag}_b": b.numpy(), f"out_{tag}_prediction": pred.tensor.detach().numpy(),
                    f"out_{tag}_loss": val.detach().numpy(), f"out_{tag}_loss_map": vmap.detach().numpy(),
                    f"out_{tag}_train_loss": train_loss.detach().numpy(), f"out_{tag}_grad_w": lm.model.w.grad.numpy(),
                    f"out_{tag}_grad_b": lm.model.b.grad.numpy()})
    np.savez_compressed(os.path.join(HERE, "r2_combined_loss.npz"), **out, meta=np.array(repr(dict(losses=conf, border=2))))
    print("wrote r2_combined_loss")


def mask_ratio(losses, lightning):
    F, Ff, Fs = 5, 7, 4
    out = {}
    # (1) mask_tensor alone on odd shapes, seeded global CPU generator
    for idx, (H, W, ratio, seed) in enumerate([(16, 16, 0.6, 5), (20, 37, 0.3, 6), (9, 64, 0.9, 7)]):
        lm = lightning.AutoRegressiveLightning.__new__(lightning.AutoRegressiveLightning)
        torch.nn.Module.__init__(lm)
        lm.mask_ratio = ratio
        x = torch.randn(2, H, W, 3, generator=torch.Generator().manual_seed(300 + idx))
        x[0, 1, 2, 0] = float("nan")   # NaN * False stays NaN; negative values give -0.0
        torch.manual_seed(seed)
        out[f"mt{idx}_x"], out[f"mt{idx}_out"] = x.numpy(), lm.mask_tensor(x).numpy()
        out[f"mt{idx}_meta"] = np.array([H, W, ratio, seed], dtype=np.float64)
    # (2) a 3-step scaled_ar rollout with mask_ratio = 0.5: one randperm per model call, in call order
    case = mg.make_case(2200, border=2)
    feat = [f"f{i}" for i in range(F)]
    cin = F + Fs + Ff
    g = torch.Generator().manual_seed(93)
    w, b = torch.randn(F, cin, 3, 3, generator=g) * 0.15, torch.randn(F, generator=g) * 0.1
    lm = new_module(lightning, case, mg.TinyConv(cin, F, w, b), "scaled_ar", mask_ratio=0.5)
    batch = ItemBatch(NamedTensor(case["inputs"].clone(), GDIMS, feat),
                      NamedTensor(case["forcing"].clone(), GDIMS, [f"g{i}" for i in range(Ff)]),
                      NamedTensor(case["outputs"].clone(), GDIMS, feat))
    torch.manual_seed(4242)
    pred, tgt = lm._common_step(batch, 0, "train")
    mask, tgt_m = lm.get_mask_on_nan(tgt)
    loss = losses.WeightedLoss("MSELoss", reduction="none")
    loss.prepare(lm, lm.interior_mask, dataset_info(lm, case, feat))
    train_loss = torch.mean(loss(pred, tgt_m, mask))
    train_loss.backward()
    out.update({f"in_{k}": v.numpy() for k, v in case.items()})
    out.update(in_w=w.numpy(), in_b=b.numpy(), out_prediction=pred.tensor.detach().numpy(),
               out_train_loss=train_loss.detach().numpy(), out_grad_w=lm.model.w.grad.numpy(), out_grad_b=lm.model.b.grad.numpy())
    np.savez_compressed(os.path.join(HERE, "r2_mask_ratio.npz"), **out,
                        meta=np.array(repr(dict(strategy="scaled_ar", K=1, T_in=1, border=2, nan=0, layout="grid", mask_ratio=0.5,
                                                seed=4242))))
    print("wrote r2_mask_ratio")


def scheduler():
    """lightning.py:442-467: AdamW(lr, betas) + get_cosine_with_min_lr_schedule_with_warmup(num_warmup_steps,
    num_training_steps=estimated_stepping_batches, min_lr=min_learning_rate), stepped once per optimizer step."""
    from transformers import get_cosine_with_min_lr_schedule_with_warmup

    out, confs = {}, []
    for idx, (lr, min_lr, warm, total, steps) in enumerate([(1e-3, 3e-7, 1000, 5000, 5200), (1e-4, 1e-6, 0, 50, 60),
                                                            (5e-4, 5e-5, 7, 20, 25), (1e-3, 1e-3, 3, 10, 12)]):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=lr, betas=(0.9, 0.95))
        sched = get_cosine_with_min_lr_schedule_with_warmup(opt, warm, total, min_lr=min_lr)
        lrs = []
        for _ in range(steps):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sched.step()
        out[f"lrs_{idx}"] = np.array(lrs, dtype=np.float64)
        confs.append(dict(lr=lr, min_lr=min_lr, warmup=warm, total=total, steps=steps))
    import transformers

    np.savez_compressed(os.path.join(HERE, "r2_scheduler.npz"), **out,
                        meta=np.array(repr(dict(confs=confs, transformers=transformers.__version__))))
    print("wrote r2_scheduler", transformers.__version__)


def ctor_signature(lightning):
    """The CLI boundary (cli.py:22-56 links dataset arguments into these parameters): names, kinds and defaults of
    AutoRegressiveLightning.__init__ (lightning.py:152-184), as data."""
    import inspect
    import json

    sig = inspect.signature(lightning.AutoRegressiveLightning.__init__)
    params = [dict(name=p.name, kind=p.kind.name, default=None if p.default is inspect._empty else repr(p.default),
                   has_default=p.default is not inspect._empty) for p in sig.parameters.values()]
    public = sorted(n for n, v in vars(lightning.AutoRegressiveLightning).items() if callable(v) and not n.startswith("_"))
    with open(os.path.join(HERE, "r2_ctor_signature.json"), "w") as f:
        json.dump(dict(params=params, public_methods=public), f, indent=1)
    print("wrote r2_ctor_signature", len(params), "parameters,", len(public), "public methods")


if __name__ == "__main__":
    losses, lightning = mg.install_stubs()
    ctor_signature(lightning)
    downscaling_only(losses, lightning)
    combined_loss(losses, lightning)
    mask_ratio(losses, lightning)
    scheduler()
