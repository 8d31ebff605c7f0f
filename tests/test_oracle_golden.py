"""
Pins the CPU oracle (oracle/) against the golden vectors produced by the unmodified
reference (tests/golden/make_golden.py).  CPU only.
"""
import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden
from oracle import losses as olosses
from oracle import rollout as orollout

FILES = golden_files()


def _tiny_model(meta, ins):
    w, b = torch.from_numpy(ins["w"]), torch.from_numpy(ins["b"])
    w.requires_grad_(True), b.requires_grad_(True)
    if meta["layout"] == "grid":
        fn = lambda x: torch.tanh(torch.nn.functional.conv2d(x, w, b, padding=1))
    else:
        fn = lambda x: torch.tanh(x @ w + b)
    return fn, w, b


def _run_oracle(meta, ins):
    t = {k: torch.from_numpy(v) for k, v in ins.items()}
    B = t["inputs"].shape[0]
    fn, w, b = _tiny_model(meta, ins)
    border, statics = t["border_mask"], t["statics"]
    inputs, forcing, outputs = t["inputs"], t["forcing"], t["outputs"]
    if meta["layout"] == "graph":
        border, statics = border.flatten(0, 1), statics.flatten(0, 1)
        inputs, forcing, outputs = inputs.flatten(2, 3), forcing.flatten(2, 3), outputs.flatten(2, 3)
    statics_b = statics.unsqueeze(0).expand(B, *statics.shape)
    pred = orollout.rollout(
        fn, inputs, forcing, outputs, statics_b, border, 1.0 - border,
        t["diff_std"], t["diff_mean"], meta["strategy"], meta["K"], bool(meta["nan"]), "train",
        features_second=(meta["layout"] == "grid"),
    )
    return pred, outputs, border, t, w, b


def test_fixtures_present():
    assert len(FILES) >= 10


@pytest.mark.parametrize("path", FILES, ids=lambda p: p.split("/")[-1][:-4])
def test_rollout_matches_reference(path):
    meta, ins, outs = load_golden(path)
    pred, *_ = _run_oracle(meta, ins)
    ref = outs["prediction"]
    got = pred.detach().numpy()
    assert got.shape == ref.shape
    # same torch ops in the same order on the same machine: bit-exact, NaNs in the same places
    np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(ref), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("path", [f for f in FILES if "grid" in f], ids=lambda p: p.split("/")[-1][:-4])
def test_losses_match_reference(path):
    meta, ins, outs = load_golden(path)
    pred, outputs, border, t, w, b = _run_oracle(meta, ins)
    interior = 1.0 - border
    mask, tgt = orollout.get_mask_on_nan(outputs, bool(meta["nan"]))
    np.testing.assert_array_equal(mask.numpy().astype(np.float32), outs["mask"])  # index/mask op: bit exact
    for tag, kind, fn in [("wmse", "mse", "w"), ("wl1", "l1", "w"), ("smse", "mse", "s"), ("sl1", "l1", "s")]:
        if fn == "w":
            wts = olosses.weighted_loss_weights(t["state_weight"], t["diff_std"], kind)
            val = olosses.weighted_loss(pred, tgt, mask, wts, interior, kind)
            vmap = olosses.weighted_loss(pred, tgt, mask, wts, interior, kind, reduce_spatial_dim=False)
            np.testing.assert_allclose(vmap.detach().numpy(), outs[f"loss_{tag}_map"], rtol=2e-6, atol=1e-7)
        else:
            val = olosses.scaled_loss(pred, tgt, mask, t["std"], interior, kind)
        np.testing.assert_allclose(val.detach().numpy(), outs[f"loss_{tag}"], rtol=2e-6, atol=1e-7)
    # training step scalar + BPTT gradient wrt the tiny model's parameters
    wts = olosses.weighted_loss_weights(t["state_weight"], t["diff_std"], "mse")
    loss = olosses.training_loss(
        pred, outputs, bool(meta["nan"]), [("WeightedLoss", 1.0, dict(weights=wts, interior_mask=interior, kind="mse"))]
    )
    loss.backward()
    np.testing.assert_allclose(loss.detach().numpy(), outs["train_loss"], rtol=2e-6)
    np.testing.assert_allclose(w.grad.numpy(), outs["grad_w"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(b.grad.numpy(), outs["grad_b"], rtol=1e-4, atol=1e-7)


def test_diff_ar_requires_single_inter_step():
    # lightning.py:688-692
    with pytest.raises(ValueError):
        orollout.strategy_params("diff_ar", 2)


def test_combined_scaled_member_raises_like_reference():
    # losses.py:304-306: (B,T) += (B,T,F) is a RuntimeError in the reference as well
    p = torch.zeros(2, 3, 4, 4, 5)
    with pytest.raises(RuntimeError):
        olosses.combined_loss(p, p, torch.ones_like(p), [("ScaledLoss", 1.0, dict(std=torch.ones(5), interior_mask=torch.ones(4, 4, 1)))])


# ------------------------------------------------------------------------------------------ rows next to the path (8f)
def _load_next(name):
    import os

    return np.load(os.path.join(os.path.dirname(__file__), "golden", name), allow_pickle=False)


def test_oracle_acc_matches_reference():
    from oracle.next_rows import acc_compute, acc_update

    z = _load_next("next_acc.npz")
    clim = torch.from_numpy(z["clim"])
    total = None
    for step in range(2):
        inc = acc_update(torch.from_numpy(z[f"pred{step}"]), torch.from_numpy(z[f"target{step}"]),
                         torch.from_numpy(z[f"mask{step}"]), clim)
        total = inc if total is None else total + inc
        np.testing.assert_allclose(total.numpy(), z[f"sum_acc{step}"], rtol=1e-6, atol=1e-7)
    names = [f"f{i}" for i in range(clim.numel())]
    res = acc_compute(total, 2.0, names, "val")
    keys = sorted(res)
    assert keys == list(z["compute_keys"])
    np.testing.assert_allclose(np.array([float(res[k]) for k in keys], dtype=np.float32), z["compute_vals"], rtol=1e-6)


def test_oracle_unnormalize_matches_reference_bit_exact():
    from oracle.next_rows import unnormalize

    z = _load_next("next_unnormalize.npz")
    out = unnormalize(torch.from_numpy(z["x"]), torch.from_numpy(z["std"]), torch.from_numpy(z["mean"]))
    assert np.array_equal(out.numpy(), z["out"])


def test_oracle_standardize_pack_matches_reference_bit_exact():
    from oracle.next_rows import standardize_pack

    z = _load_next("next_pack.npz")
    full = standardize_pack(z["raw"], z["mean"], z["std"])
    assert np.array_equal(full[:, :1], z["inputs"]) and np.array_equal(full[:, 1:], z["outputs"])


def _stats_from_moments(z, moments_fn):
    """the reference's loops (compute_dataset_stats.py:11-127) on top of per-batch NaN-aware moments"""
    names = {"inputs": ["a", "b", "c"], "outputs": ["a", "b", "c"], "forcing": ["f0", "f1"]}
    res = {}
    for kind in ("inputs", "outputs", "forcing"):
        F = len(names[kind])
        first = torch.from_numpy(z[f"batch0__{kind}"])
        m0 = moments_fn(first.reshape(1, -1, F))
        best_min, best_max = m0[3, 0], m0[4, 0]
        s1 = torch.zeros(F); s2 = torch.zeros(F); counter = 0
        for i in range(3):
            t = torch.from_numpy(z[f"batch{i}__{kind}"])
            m = moments_fn(t)
            counter += t.shape[0]
            s1 += torch.nansum(m[0] / m[2], dim=0); s2 += torch.nansum(m[1] / m[2], dim=0)
            best_min = torch.minimum(best_min, m[3, 0]); best_max = torch.maximum(best_max, m[4, 0])
        mean = s1 / counter
        std = torch.sqrt(s2 / counter - mean**2)
        for j, n in enumerate(names[kind]):
            res[f"{kind}__{n}"] = dict(mean=mean[j], std=std[j], min=best_min[j], max=best_max[j])
    s1 = torch.zeros(3); s2 = torch.zeros(3); counter = 0
    for i in range(3):
        io = torch.cat([torch.from_numpy(z[f"batch{i}__inputs"]), torch.from_numpy(z[f"batch{i}__outputs"])], dim=1)
        m = moments_fn(io[:, :-1], io[:, 1:])
        counter += io.shape[0]
        s1 += torch.nansum(m[0] / m[2], dim=0); s2 += torch.nansum(m[1] / m[2], dim=0)
    dm = s1 / counter
    ds = torch.sqrt(s2 / counter - dm**2)
    for j, n in enumerate(names["inputs"]):
        res[f"diff__{n}"] = dict(mean=dm[j], std=ds[j])
    return res


def test_oracle_dataset_stats_match_reference():
    from oracle.next_rows import nan_moments

    z = _load_next("next_stats.npz")
    res = _stats_from_moments(z, nan_moments)
    for key, d in res.items():
        for stat, v in d.items():
            np.testing.assert_allclose(float(v), float(z[f"{key}__{stat}"]), rtol=2e-5, atol=2e-6, err_msg=f"{key} {stat}")
