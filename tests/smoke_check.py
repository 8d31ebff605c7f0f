"""
smoke(): one tiny autoregressive training step (2 AR steps, HalfUNet, scaled_ar, weighted MSE, backward, AdamW)
through the HIP path on a GPU, checked against the CPU oracle (oracle/ -- test infrastructure, used here only as
the checker).  Called by __graft_entry__.smoke().
"""

import torch


def run(device: torch.device) -> None:
    from oracle import losses as olosses
    from oracle import rollout as orollout
    from oracle.halfunet import HalfUNetRef

    from py4cast_amd import _lib as L
    from py4cast_amd.base import DatasetInfo, ItemBatch, Statics, Stats
    from py4cast_amd.lightning import AutoRegressiveLightning
    from py4cast_amd.namedtensor import NamedTensor

    L.lib()  # fails loudly if the extension is missing
    B, T, H, W, F, Ff, Fs = 2, 2, 32, 32, 6, 5, 4
    g = torch.Generator().manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g)
    ru = lambda *s: torch.rand(*s, generator=g)
    case = dict(inputs=rn(B, 1, H, W, F), forcing=ru(B, T, H, W, Ff), outputs=rn(B, T, H, W, F), statics=ru(H, W, Fs),
                diff_std=ru(F) + 0.5, diff_mean=rn(F) * 0.01, std=ru(F) + 0.5, state_weight=1.0 + ru(F))
    bm = torch.zeros(H, W, 1)
    bm[:2], bm[-2:], bm[:, :2], bm[:, -2:] = 1, 1, 1, 1
    case["statics"][..., 3:4] = bm
    names = [f"f{i}" for i in range(F)]
    gs = NamedTensor(case["statics"], ["lat", "lon", "features"], ["x", "y", "geopotential", "border_mask"])
    info = DatasetInfo(
        "smoke", Statics(gs, (H, W)), Stats({n: {"std": case["std"][i], "mean": torch.tensor(0.0)} for i, n in enumerate(names)}),
        Stats({n: {"std": case["diff_std"][i], "mean": case["diff_mean"][i]} for i, n in enumerate(names)}),
        {n: float(case["state_weight"][i]) for i, n in enumerate(names)}, {"input_output": names}, F, Ff)
    torch.manual_seed(0)
    lm = AutoRegressiveLightning(
        {}, info, None, num_pred_steps_train=T, batch_size=B, model_name="HalfUNet",
        losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
        training_strategy="scaled_ar", learning_rate=1e-3, betas=(0.9, 0.95))
    ref = HalfUNetRef(F + Fs + Ff, F).double()
    ref.load_state_dict(lm.model.state_dict())
    lm = lm.to(device).train()
    dims = ["batch", "timestep", "lat", "lon", "features"]
    batch = ItemBatch(NamedTensor(case["inputs"].to(device), dims, names),
                      NamedTensor(case["forcing"].to(device), dims, [f"g{i}" for i in range(Ff)]),
                      NamedTensor(case["outputs"].to(device), dims, names))
    opt = lm.configure_optimizers()["optimizer"]
    loss = lm.training_step(batch, 0)
    loss.backward()
    gnorm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in lm.model.parameters())))
    opt.step()
    torch.cuda.synchronize(device)

    c = {k: v.double() for k, v in case.items()}
    interior = 1.0 - bm.double()
    ref.train()
    pred = orollout.rollout(ref, c["inputs"], c["forcing"], c["outputs"], c["statics"].unsqueeze(0).expand(B, H, W, Fs),
                            bm.double(), interior, c["diff_std"], c["diff_mean"], "scaled_ar", 1, False, "train", features_second=True)
    wts = olosses.weighted_loss_weights(c["state_weight"], c["diff_std"], "mse")
    lref = olosses.training_loss(pred, c["outputs"], False, [("WeightedLoss", 1.0, dict(weights=wts, interior_mask=interior, kind="mse"))])
    lref.backward()
    gref = float(torch.sqrt(sum((p.grad ** 2).sum() for p in ref.parameters())))
    rel = abs(loss.item() - lref.item()) / abs(lref.item())
    assert rel < 1e-4, f"smoke: loss {loss.item()} vs oracle {lref.item()} (rel {rel:.2e})"
    assert abs(gnorm - gref) / gref < 5e-2, f"smoke: grad norm {gnorm} vs oracle {gref}"
    print(f"smoke ok: loss {loss.item():.6f} (oracle {lref.item():.6f}, rel {rel:.1e}), |grad| {gnorm:.4f} (oracle {gref:.4f})")
