#!/usr/bin/env python3
"""
Golden-vector generator -- runs ONLY in the build container (needs /root/reference).

It executes the *unmodified* reference files ``py4cast/losses.py`` and
``py4cast/lightning.py`` on CPU through ``sys.modules`` stubs of their non-hot-path
dependencies (recipe: SURVEY.md appendix B) and dumps seeded inputs + the reference's
outputs / gradients as small ``.npz`` fixtures next to this script.  Everything stubbed
is metadata plumbing (NamedTensor shim, Lightning base class, ...); every arithmetic
op executed on the fixtures belongs to the reference.

    python tests/golden/make_golden.py            # regenerates tests/golden/*.npz

The fixtures are data only; the reference never travels to the GPU box.
"""

import dataclasses
import importlib
import itertools
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("PY4CAST_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- stubs
def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class NamedTensor:  # metadata-only shim of mfai.pytorch.namedtensor.NamedTensor
    SPATIAL = ("lat", "lon", "ngrid")

    def __init__(self, tensor, names, feature_names, feature_dim_name="features"):
        assert tensor.dim() == len(names)
        assert tensor.shape[names.index(feature_dim_name)] == len(feature_names)
        self.tensor, self.names, self.feature_names = tensor, list(names), list(feature_names)
        self.feature_names_to_idx = {n: i for i, n in enumerate(feature_names)}

    device = property(lambda s: s.tensor.device)
    spatial_dim_idx = property(lambda s: sorted(s.names.index(n) for n in set(s.SPATIAL) & set(s.names)))
    num_spatial_dims = property(lambda s: len(s.spatial_dim_idx))

    def dim_size(self, n):
        return self.tensor.size(self.names.index(n))

    def dim_index(self, n):
        return self.names.index(n)

    def select_tensor_dim(self, n, i):
        return self.tensor.select(self.names.index(n), i)

    def select_dim(self, n, i):
        return NamedTensor(self.select_tensor_dim(n, i), [x for x in self.names if x != n], self.feature_names)

    def index_select_tensor_dim(self, n, idx):
        return self.tensor.index_select(
            self.names.index(n), torch.tensor(list(idx), dtype=torch.int64, device=self.device)
        )

    @staticmethod
    def new_like(t, o):
        return NamedTensor(t, o.names.copy(), o.feature_names.copy())

    def clone(self):
        return NamedTensor(self.tensor.clone(), self.names.copy(), self.feature_names.copy())

    def flatten_(self, name, s, e):
        self.tensor = torch.flatten(self.tensor, s, e)
        self.names = self.names[:s] + [name] + self.names[e + 1 :]


@dataclasses.dataclass
class ItemBatch:  # mirrors base.py:147-170
    inputs: NamedTensor
    forcing: NamedTensor
    outputs: NamedTensor
    batch_size = property(lambda s: s.outputs.dim_size("batch"))
    num_input_steps = property(lambda s: s.inputs.dim_size("timestep"))
    num_pred_steps = property(lambda s: s.outputs.dim_size("timestep"))


class ModelType:
    GRAPH = 1
    CONVOLUTIONAL = 2
    VISION_TRANSFORMER = 3


class ModelABC:
    def check_required_attributes(self):
        pass


def install_stubs():
    sys.path.insert(0, REF)

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            self.hparams = {}

    class LightningDataModule:
        def __init__(self, *a, **k):
            pass

    L = stub("lightning", LightningModule=LightningModule, LightningDataModule=LightningDataModule)
    L.pytorch = stub("lightning.pytorch", LightningModule=LightningModule)
    stub("lightning.pytorch.loggers", MLFlowLogger=type("MLFlowLogger", (), {}))
    stub("lightning.pytorch.utilities", rank_zero_only=lambda f: f)
    stub("dataclasses_json", dataclass_json=lambda c: c)
    for n in ("mlflow", "mlflow.pytorch", "mlflow.models", "mlflow.models.signature", "torchinfo", "gif"):
        stub(n)
    sys.modules["mlflow.models.signature"].infer_signature = lambda *a, **k: None
    sys.modules["torchinfo"].summary = lambda *a, **k: None
    sys.modules["gif"].frame = lambda f: f
    stub("mfai")
    stub("mfai.pytorch")
    stub("mfai.pytorch.namedtensor", NamedTensor=NamedTensor)
    stub("mfai.pytorch.losses")
    stub("mfai.pytorch.losses.perceptual", PerceptualLoss=object)
    stub("mfai.pytorch.models", registry={"PanguWeather": None, "ArchesWeather": None})
    stub("mfai.pytorch.models.base", ModelType=ModelType, ModelABC=ModelABC)
    stub(
        "mfai.pytorch.models.utils",
        expand_to_batch=lambda t, b: t.unsqueeze(0).expand(b, *t.shape),
        features_last_to_second=lambda x: x.movedim(-1, 1),
        features_second_to_last=lambda x: x.movedim(1, -1),
    )
    stub("py4cast.datasets", get_datasets=None)
    stub("py4cast.datasets.base", DatasetInfo=object, ItemBatch=ItemBatch, NamedTensor=NamedTensor, Statics=object)
    stub("py4cast.io")
    stub("py4cast.io.outputs", OutputSavingSettings=None, save_gifs=None, save_named_tensors_to_grib=None)
    stub("py4cast.metrics", MetricACC=None, MetricPSDK=None, MetricPSDVar=None)
    stub(
        "py4cast.plots",
        PredictionEpochPlot=None,
        PredictionTimestepPlot=None,
        SpatialErrorPlot=None,
        StateErrorPlot=None,
    )
    stub("py4cast.utils", str_to_dtype={"32-true": torch.float32, "bf16-true": torch.bfloat16})
    losses = importlib.import_module("py4cast.losses")  # unmodified reference file
    lightning = importlib.import_module("py4cast.lightning")  # unmodified reference file
    return losses, lightning


# --------------------------------------------------------------------------- fixtures
class StatsLike:
    """Stats.to_list contract (access.py:368-390): name -> {stat: 0-d tensor}."""

    def __init__(self, d):
        self.stats = d

    def __getitem__(self, k):
        return self.stats[k]

    def to_list(self, stat_name, shortnames, dtype=torch.float32):
        return torch.stack([self[n][stat_name] for n in shortnames], dim=0).type(dtype)


class TinyConv(torch.nn.Module):
    """Deterministic 3x3 conv + tanh, NCHW (features_second) -- weights are part of the fixture."""

    model_type = ModelType.CONVOLUTIONAL
    features_second = True

    def __init__(self, cin, cout, w, b):
        super().__init__()
        self.w = torch.nn.Parameter(w.clone())
        self.b = torch.nn.Parameter(b.clone())

    def forward(self, x):
        return torch.tanh(torch.nn.functional.conv2d(x, self.w, self.b, padding=1))


class TinyLinear(torch.nn.Module):
    """Per-node linear + tanh on (B,N,C) -- graph layout (features last, 1 spatial dim)."""

    model_type = ModelType.GRAPH
    features_second = False

    def __init__(self, cin, cout, w, b):
        super().__init__()
        self.w = torch.nn.Parameter(w.clone())
        self.b = torch.nn.Parameter(b.clone())

    def forward(self, x):
        return torch.tanh(x @ self.w + self.b)


def make_case(seed, B=2, T=3, T_in=1, H=16, W=16, F=5, Ff=7, Fs=4, border=0, nan=False):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    ru = lambda *s: torch.rand(*s, generator=g)
    d = dict(
        inputs=rn(B, T_in, H, W, F).clamp(-3, 3),
        forcing=ru(B, T, H, W, Ff),
        outputs=rn(B, T, H, W, F).clamp(-3, 3),
        statics=ru(H, W, Fs),
        diff_std=ru(F) + 0.5,
        diff_mean=rn(F) * 0.01,
        std=ru(F) + 0.5,
        state_weight=1.0 + ru(F),
    )
    bm = torch.zeros(H, W, 1)
    if border > 0:
        bm[:border], bm[-border:], bm[:, :border], bm[:, -border:] = 1, 1, 1, 1
    d["border_mask"] = bm
    d["statics"][..., 3:4] = bm
    if nan:
        # NaNs in inputs / forcing / targets, incl. one pixel that is NaN for every (b,t,f)
        d["inputs"][0, 0, 3, 4, 1] = float("nan")
        d["forcing"][1, 1, 5, 6, 2] = float("nan")
        d["outputs"][:, :, 7, 8, :] = float("nan")
        d["outputs"][0, 1, 2, 2, 0] = float("nan")
    return d


def run_reference(losses, lightning, case, strategy, K, layout, mask_on_nan, loss_specs, w, b):
    B, T_in, H, W, F = case["inputs"].shape
    T = case["outputs"].shape[1]
    Ff, Fs = case["forcing"].shape[-1], case["statics"].shape[-1]
    feat = [f"f{i}" for i in range(F)]
    gdims = ["batch", "timestep", "lat", "lon", "features"]
    lm = lightning.AutoRegressiveLightning.__new__(lightning.AutoRegressiveLightning)
    torch.nn.Module.__init__(lm)
    cin = T_in * F + Fs + Ff + int(mask_on_nan)
    lm.model = (TinyConv if layout == "grid" else TinyLinear)(cin, F, w, b)
    lm.training_strategy, lm.num_inter_steps = strategy, K
    lm.channels_last, lm.mask_ratio, lm.mask_on_nan = False, 0, mask_on_nan
    lm.diff_stats = StatsLike({n: {"std": case["diff_std"][i], "mean": case["diff_mean"][i]} for i, n in enumerate(feat)})
    lm.stats = StatsLike({n: {"std": case["std"][i]} for i, n in enumerate(feat)})
    bm, st = case["border_mask"].clone(), case["statics"].clone()
    if layout == "graph":
        bm, st = bm.flatten(0, 1), st.flatten(0, 1)
    lm.register_buffer("border_mask", bm)
    lm.register_buffer("interior_mask", 1.0 - bm)
    lm.register_buffer("grid_static_features", st.unsqueeze(0).expand(B, *st.shape).clone())
    batch = ItemBatch(
        NamedTensor(case["inputs"].clone(), gdims, feat),
        NamedTensor(case["forcing"].clone(), gdims, [f"g{i}" for i in range(Ff)]),
        NamedTensor(case["outputs"].clone(), gdims, feat),
    )
    pred, tgt = lm._common_step(batch, 0, "train")
    out = {"prediction": pred.tensor.detach().numpy().copy()}
    mask, tgt_masked = lm.get_mask_on_nan(tgt)
    out["mask"] = mask.numpy().astype(np.float32)

    class DI:
        state_weights = {n: float(case["state_weight"][i]) for i, n in enumerate(feat)}
        stats = lm.stats
        diff_stats = lm.diff_stats

    if layout == "graph":
        return out  # reference losses raise IndexError on graph layout (losses.py:156,197)
    for tag, cls, kind in loss_specs:
        lossobj = getattr(losses, cls)(kind, reduction="none")
        lossobj.prepare(lm, lm.interior_mask, DI)
        val = lossobj(pred, tgt_masked, mask)
        out[f"loss_{tag}"] = val.detach().numpy().copy()
        if cls == "WeightedLoss":
            out[f"loss_{tag}_map"] = lossobj(pred, tgt_masked, mask, reduce_spatial_dim=False).detach().numpy().copy()
            if tag == "wmse":  # training-step scalar + gradient wrt the model parameters (BPTT)
                lm.model.zero_grad()
                batch_loss = torch.mean(val)
                batch_loss.backward()
                out["train_loss"] = batch_loss.detach().numpy().copy()
                out["grad_w"] = lm.model.w.grad.numpy().copy()
                out["grad_b"] = lm.model.b.grad.numpy().copy()
    return out


LOSS_SPECS = [
    ("wmse", "WeightedLoss", "MSELoss"),
    ("wl1", "WeightedLoss", "L1Loss"),
    ("smse", "ScaledLoss", "MSELoss"),
    ("sl1", "ScaledLoss", "L1Loss"),
]


def main():
    losses, lightning = install_stubs()
    cases = []
    for strategy, K, T_in, border, nan, layout in [
        ("scaled_ar", 1, 1, 0, False, "grid"),
        ("scaled_ar", 1, 1, 2, False, "grid"),
        ("scaled_ar", 2, 1, 2, False, "grid"),
        ("scaled_ar", 1, 2, 2, False, "grid"),
        ("diff_ar", 1, 1, 0, False, "grid"),
        ("diff_ar", 1, 2, 2, False, "grid"),
        ("scaled_ar", 1, 1, 2, True, "grid"),
        ("diff_ar", 1, 2, 0, True, "grid"),
        ("scaled_ar", 1, 1, 2, False, "graph"),
        ("diff_ar", 1, 2, 0, False, "graph"),
    ]:
        cases.append((strategy, K, T_in, border, nan, layout))
    for idx, (strategy, K, T_in, border, nan, layout) in enumerate(cases):
        case = make_case(1000 + idx, T_in=T_in, border=border, nan=nan)
        F, Ff, Fs = 5, 7, 4
        cin = T_in * F + Fs + Ff + int(nan)
        g = torch.Generator().manual_seed(77 + idx)
        if layout == "grid":
            w = torch.randn(F, cin, 3, 3, generator=g) * 0.15
        else:
            w = torch.randn(cin, F, generator=g) * 0.3
        b = torch.randn(F, generator=g) * 0.1
        out = run_reference(losses, lightning, case, strategy, K, layout, nan, LOSS_SPECS, w, b)
        name = f"rollout_{idx:02d}_{strategy}_K{K}_Tin{T_in}_b{border}_{'nan' if nan else 'nonan'}_{layout}"
        meta = dict(strategy=strategy, K=K, T_in=T_in, border=border, nan=int(nan), layout=layout)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            **{f"in_{k}": v.numpy() for k, v in case.items()},
            in_w=w.numpy(),
            in_b=b.numpy(),
            **{f"out_{k}": v for k, v in out.items()},
            meta=np.array(repr(meta)),
        )
        print("wrote", name, {k: v.shape for k, v in out.items()})

    # error behaviour fixtures (documented, asserted by tests): diff_ar with K=2 -> ValueError
    lm = lightning.AutoRegressiveLightning.__new__(lightning.AutoRegressiveLightning)
    torch.nn.Module.__init__(lm)
    lm.training_strategy, lm.num_inter_steps = "diff_ar", 2
    try:
        lm._strategy_params()
        raise SystemExit("expected ValueError")
    except ValueError as e:
        print("diff_ar K=2 ->", type(e).__name__, e)


if __name__ == "__main__":
    main()
