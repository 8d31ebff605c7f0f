"""GPU parity of the rows next to the hot path: kernels vs the oracle and the reference's golden vectors."""

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_unnormalize_bit_exact_vs_reference(gpu_device):
    from py4cast_amd import ops

    z = np.load(os.path.join(GOLD, "next_unnormalize.npz"))
    x = torch.from_numpy(z["x"]).to(gpu_device)
    out = ops.unnormalize(x, torch.from_numpy(z["std"]).to(gpu_device), torch.from_numpy(z["mean"]).to(gpu_device))
    assert np.array_equal(out.cpu().numpy(), z["out"])  # two rounded steps, no FMA
    # in place, benchmark-sized feature count, odd row count
    g = torch.Generator().manual_seed(1)
    y = torch.randn(3, 2, 33, 17, 60, generator=g)
    std, mean = torch.rand(60, generator=g) + 0.5, torch.randn(60, generator=g)
    ref = y.clone(); ref *= std; ref += mean
    yd = y.to(gpu_device)
    r = ops.unnormalize(yd, std.to(gpu_device), mean.to(gpu_device), out=yd)
    assert r.data_ptr() == yd.data_ptr() and torch.equal(yd.cpu(), ref)


def test_pack_standardize_bit_exact_vs_reference(gpu_device):
    from py4cast_amd import ops
    from py4cast_amd.datapipe import load_batch
    from py4cast_amd.namedtensor import NamedTensor

    z = np.load(os.path.join(GOLD, "next_pack.npz"))
    raw = torch.from_numpy(z["raw"]).to(gpu_device)
    full = ops.pack_standardize(raw, torch.from_numpy(z["mean"]).to(gpu_device), torch.from_numpy(z["std"]).to(gpu_device))
    assert np.array_equal(full[:, :1].cpu().numpy(), z["inputs"]) and np.array_equal(full[:, 1:].cpu().numpy(), z["outputs"])

    class Stats:
        def to_list(self, stat, names, dtype=torch.float32):
            return torch.from_numpy(z[stat]).type(dtype)

    names = [f"p{i}" for i in range(raw.shape[0])]
    forcing = NamedTensor(torch.zeros(2, 3, 6, 5, 1, device=gpu_device), ["batch", "timestep", "lat", "lon", "features"], ["x"])
    batch = load_batch(raw, names, forcing, Stats(), num_input_steps=1)
    assert batch.num_input_steps == 1 and batch.num_pred_steps == 3 and batch.batch_size == 2
    assert np.array_equal(batch.outputs.tensor.cpu().numpy(), z["outputs"])
    # ragged sizes: rows not a multiple of 64, features not a multiple of 32
    g = torch.Generator().manual_seed(2)
    r2 = torch.randn(37, 2, 1, 9, 11, generator=g) * 7 + 3
    m2, s2 = torch.randn(37, generator=g), torch.rand(37, generator=g) + 0.3
    ref = torch.stack([(r2[f] - m2[f]) / s2[f] for f in range(37)], dim=-1)
    out = ops.pack_standardize(r2.to(gpu_device), m2.to(gpu_device), s2.to(gpu_device))
    assert torch.equal(out.cpu(), ref)


def test_metric_acc_matches_reference(gpu_device):
    from py4cast_amd.metrics import MetricACC
    from py4cast_amd.namedtensor import NamedTensor

    z = np.load(os.path.join(GOLD, "next_acc.npz"))
    F = z["clim"].shape[0]
    names = [f"f{i}" for i in range(F)]

    class Stats:
        def to_list(self, stat, ns, dtype=torch.float32):
            return torch.from_numpy(z["clim"]).type(dtype)

    class Info:
        shortnames = {"input_output": names[:3], "output": names[3:]}
        stats = Stats()

    with pytest.warns(UserWarning):
        m = MetricACC(Info())
    dims = ["batch", "timestep", "lat", "lon", "features"]
    for step in range(2):
        p = NamedTensor(torch.from_numpy(z[f"pred{step}"]).to(gpu_device), dims, names)
        t = NamedTensor(torch.from_numpy(z[f"target{step}"]).to(gpu_device), dims, names)
        m.update(p, t, torch.from_numpy(z[f"mask{step}"]).to(gpu_device))
        np.testing.assert_allclose(m.sum_acc.cpu().numpy(), z[f"sum_acc{step}"], rtol=2e-5, atol=2e-6)
    res = m.compute(prefix="val")
    keys = sorted(res)
    assert keys == list(z["compute_keys"])
    np.testing.assert_allclose(np.array([float(res[k]) for k in keys], dtype=np.float32), z["compute_vals"], rtol=2e-5, atol=2e-6)
    assert m.step_count == 0  # compute() resets, as the reference


def test_acc_sums_full_size_properties(gpu_device):
    """At the benchmark size: ACC(x, x) = 1 for every (t, f); shifting both by the climate mean leaves it unchanged."""
    from py4cast_amd import ops

    g = torch.Generator(device="cpu").manual_seed(3)
    B, T, H, W, F = 1, 1, 512, 512, 60
    x = torch.randn(B, T, H, W, F, generator=g).to(gpu_device)
    clim = torch.randn(F, generator=g).to(gpu_device)
    s = ops.acc_sums(x, x, ops.MaskSpec(0), clim)
    acc = s[0] / torch.sqrt(s[1] * s[2])
    assert torch.allclose(acc, torch.ones_like(acc), atol=1e-5)
    y = torch.randn(B, T, H, W, F, generator=g).to(gpu_device)
    a1 = ops.acc_sums(x + clim, y + clim, ops.MaskSpec(0), clim)
    a0 = ops.acc_sums(x, y, ops.MaskSpec(0), torch.zeros_like(clim))
    assert torch.allclose(a1, a0, rtol=1e-4, atol=1e-6)


def test_dataset_stats_match_reference(gpu_device):
    """compute_mean_std_min_max / compute_time_step_stats mirrors (one p4c_nan_moments pass per batch) against the
    statistics the unmodified reference computed on the same three batches (two of them with NaNs)."""
    import types

    from py4cast_amd import dataset_stats as ds
    from py4cast_amd.namedtensor import NamedTensor

    z = np.load(os.path.join(GOLD, "next_stats.npz"))
    names = {"inputs": ["a", "b", "c"], "outputs": ["a", "b", "c"], "forcing": ["f0", "f1"]}
    dims = ["batch", "timestep", "lat", "lon", "features"]
    batches = [types.SimpleNamespace(**{k: NamedTensor(torch.from_numpy(z[f"batch{i}__{k}"]), dims, names[k]) for k in names})
               for i in range(3)]

    class DS:
        def __init__(self, standardize):
            self.settings = types.SimpleNamespace(standardize=standardize)

        def torch_dataloader(self):
            return list(batches)

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for kind in names:
            st = ds.compute_mean_std_min_max(DS(False), kind, gpu_device)
            for n, d in st.items():
                for k, v in d.items():
                    np.testing.assert_allclose(float(v), float(z[f"{kind}__{n}__{k}"]), rtol=2e-5, atol=2e-6, err_msg=f"{kind} {n} {k}")
    diff = ds.compute_time_step_stats(DS(True), gpu_device)
    for n in names["inputs"]:
        for k in ("mean", "std"):
            np.testing.assert_allclose(float(diff[n][k]), float(z[f"diff__{n}__{k}"]), rtol=2e-5, atol=2e-6)
    assert int(diff["f0"]["mean"]) == 0 and int(diff["f1"]["std"]) == 1
    with pytest.raises(ValueError):
        ds.compute_mean_std_min_max(DS(True), "inputs", gpu_device)
    # kernel vs oracle on a ragged, NaN-laden tensor incl. an all-NaN (sample, feature) column
    from oracle.next_rows import nan_moments as ref

    g = torch.Generator().manual_seed(8)
    x = torch.randn(3, 2, 7, 9, 5, generator=g)
    x[torch.rand(x.shape, generator=g) < 0.1] = float("nan")
    x[1, ..., 2] = float("nan")
    got, exp = ds.nan_moments(x.to(gpu_device)).cpu(), ref(x)
    assert torch.equal(got[2], exp[2]) and torch.equal(got[3], exp[3]) and torch.equal(got[4], exp[4])
    np.testing.assert_allclose(got[:2].numpy(), exp[:2].numpy(), rtol=1e-5, atol=1e-5)


def test_titan_npy_layout_to_device_batch_bit_exact(gpu_device, tmp_path):
    """8f-4: planes written in Titan's layout (one .npy per date and parameter) -> pinned buffer -> one transfer -> one
    standardise + pack kernel, against the reference's np.load / (arr - mean) / std / stack / concat (datasets/base.py:431-527)."""
    import datetime as dt

    from py4cast_amd import diskio
    from py4cast_amd.base import Stats
    from py4cast_amd.namedtensor import NamedTensor

    rng = np.random.default_rng(3)
    H, W, B, T, T_in = 12, 20, 2, 4, 1
    params = [("aro_t2m", 2, "heightAboveGround"), ("aro_z", 500, "isobaricInhPa"), ("aro_u", 850, "isobaricInhPa")]
    dates = [[dt.datetime(2023, 1, 1 + b, 3 * t) for t in range(T)] for b in range(B)]
    names, planes = [], {}
    for (n, lv, lt) in params:
        names.append(f"{n}_{lv}{'m' if lt == 'heightAboveGround' else 'hpa'}")
        for b in range(B):
            for t in range(T):
                p = diskio.titan_plane_path(tmp_path, n, lv, lt, dates[b][t])
                p.parent.mkdir(parents=True, exist_ok=True)
                arr = (rng.standard_normal((H, W)) * 50 + 270).astype(np.float32)
                np.save(p, arr)
                planes[(names[-1], b, t)] = arr
    stats_dict = {nm: {"mean": torch.tensor(float(260 + 5 * i)), "std": torch.tensor(float(3 + i))} for i, nm in enumerate(names)}
    diskio.save_stats(stats_dict, tmp_path / "parameters_stats.pt")
    stats = diskio.load_stats(tmp_path / "parameters_stats.pt")
    forcing = NamedTensor(torch.zeros(B, T - T_in, H, W, 1, device=gpu_device), ["batch", "timestep", "lat", "lon", "features"], ["x"])
    batch = diskio.load_titan_batch(tmp_path, params, dates, stats, T_in, forcing, device=gpu_device)
    # the reference's arithmetic, parameter by parameter (numpy float32 planes, float32 0-d statistics)
    per_param = []
    for nm in names:
        arr = np.stack([np.stack([planes[(nm, b, t)] for t in range(T)]) for b in range(B)])          # (B,T,H,W)
        mean, std = np.asarray(stats[nm]["mean"]), np.asarray(stats[nm]["std"])
        per_param.append(torch.from_numpy((arr - mean) / std))
    ref = torch.stack(per_param, dim=-1).float()
    assert batch.inputs.tensor.shape == (B, T_in, H, W, 3) and batch.outputs.tensor.shape == (B, T - T_in, H, W, 3)
    assert torch.equal(batch.inputs.tensor.cpu(), ref[:, :T_in]) and torch.equal(batch.outputs.tensor.cpu(), ref[:, T_in:])
    assert batch.inputs.feature_names == names


def test_output_staging_planes_bit_exact_and_overlapped(gpu_device):
    """8f-3, second half: the un-normalised prediction leaves for the writers as feature-major planes in pinned host memory
    (py4cast_amd.outputs.OutputStager).  Bit-exact with the reference's per-feature passes (lightning.py:1162-1169, golden vector of
    the unmodified lines) and with what the reference's writers read (`tensor[:, :, idx].cpu().numpy()`, io/outputs.py:193-197);
    two batches in flight do not disturb each other; ragged spatial sizes (N not a multiple of the 64-point tile)."""
    from py4cast_amd import ops
    from py4cast_amd.namedtensor import NamedTensor
    from py4cast_amd.outputs import OutputStager

    z = np.load(os.path.join(GOLD, "next_unnormalize.npz"))
    x = torch.from_numpy(z["x"]).to(gpu_device)                    # (B,T,H,W,F)
    std, mean = torch.from_numpy(z["std"]).to(gpu_device), torch.from_numpy(z["mean"]).to(gpu_device)
    planes = ops.unnormalize_planes(x, std, mean)
    want = np.moveaxis(z["out"], -1, 2)                             # (B,T,F,H,W)
    assert planes.shape == want.shape and np.array_equal(planes.cpu().numpy(), want)

    g = torch.Generator().manual_seed(2)
    names = ["batch", "timestep", "lat", "lon", "features"]
    stager = OutputStager(gpu_device)
    batches, refs = [], []
    for i in range(3):
        y = torch.randn(2, 3, 33, 17, 60, generator=g)
        s, m = torch.rand(60, generator=g) + 0.5, torch.randn(60, generator=g)
        ref = y.clone(); ref *= s; ref += m
        batches.append((NamedTensor(y.to(gpu_device), names, [f"f{k}" for k in range(60)]), s.to(gpu_device), m.to(gpu_device)))
        refs.append(ref)
    s0 = stager.submit(*batches[0])
    s1 = stager.submit(*batches[1])                                  # two batches in flight, two slots
    for slot, ref in ((s0, refs[0]), (s1, refs[1])):
        staged = stager.wait(slot)
        assert staged.planes.shape == (2, 3, 60, 33, 17)
        for (b, t, f) in ((0, 0, 0), (1, 2, 59), (0, 1, 17)):
            assert np.array_equal(staged.plane(b, t, f"f{f}"), ref[b, t, :, :, f].numpy())    # the writers' (lat, lon) plane
        assert np.array_equal(staged.planes, np.moveaxis(ref.numpy(), -1, 2))
    s2 = stager.submit(*batches[2])                                  # reuses slot 0 after its copy has completed
    assert s2 == s0 and np.array_equal(stager.wait(s2).planes, np.moveaxis(refs[2].numpy(), -1, 2))
    # the loop form: writers of batch i run while batch i+1 is on the device
    seen = []
    stager.run(range(3), lambda b, i: batches[b][0], batches[0][1], batches[0][2], lambda staged, i: seen.append((i, staged.planes[0, 0, 0, 0, 0])))
    assert [i for i, _ in seen] == [0, 1, 2]


def test_predict_step_stages_outputs(gpu_device):
    """predict_step with an OutputStager attached: returns the un-normalised NamedTensor as before AND the same numbers arrive on
    the host as planes."""
    from helpers import make_batch, make_dataset_info, register_test_models, synthetic_case
    from py4cast_amd.lightning import AutoRegressiveLightning
    from py4cast_amd.outputs import OutputStager

    register_test_models()
    case = synthetic_case(seed=8, B=2, T=2, H=16, W=16, F=5, Ff=7, Fs=4, border=0)
    info = make_dataset_info(case, 7)
    lm = AutoRegressiveLightning({}, info, None, num_pred_steps_train=2, num_pred_steps_val_test=2, batch_size=2, model_name="TinyConvModel",
                                 training_strategy="scaled_ar").to(gpu_device)
    with torch.no_grad():
        lm.model.w.normal_(0, 0.1)
    lm.training_step(make_batch(case, gpu_device), 0)          # records the feature names (lightning.py:541-545)
    lm.output_stager = OutputStager(gpu_device)
    preds = lm.predict_step(make_batch(case, gpu_device), 0)
    staged = lm.output_stager.wait(lm.staged_slot)
    assert np.array_equal(staged.planes, np.moveaxis(preds.tensor.cpu().numpy(), -1, 2))
    assert staged.feature_names == preds.feature_names
