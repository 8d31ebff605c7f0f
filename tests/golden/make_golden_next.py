#!/usr/bin/env python3
"""
Golden vectors for the rows next to the hot path -- runs ONLY in the build container.

Executes the reference's own code on seeded inputs through ``sys.modules`` stubs of its non-arithmetic
dependencies and stores inputs + outputs as ``next_*.npz`` next to this script:

* ``MetricACC.update / compute``            (py4cast/metrics.py:355-455, unmodified file)
* the un-normalise loop of ``predict_step`` (py4cast/lightning.py:1162-1169, the source lines are taken from the
  imported function with ``inspect`` and executed as they stand)
* ``Sample.get_param_tensor`` standardisation + ``collate_fn`` (py4cast/datasets/base.py:431-453, 173-195,
  unmodified file; accessor / NamedTensor are metadata stubs)

    python tests/golden/make_golden_next.py
"""

import importlib
import inspect
import os
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stub helpers, NamedTensor shim)


class NT(mg.NamedTensor):
    """adds the two constructors collate / load use (metadata only)"""

    @staticmethod
    def concat(nts):
        return NT(torch.cat([n.tensor for n in nts], dim=-1), nts[0].names.copy(), [f for n in nts for f in n.feature_names])

    @staticmethod
    def expand_to_batch_like(tensor, other):
        return NT(tensor, ["batch"] + other.names, other.feature_names.copy())


def gen_acc():
    class Metric(torch.nn.Module):  # torchmetrics.Metric: state registry only
        def __init__(self):
            super().__init__()
            self._defaults = {}

        device = property(lambda s: torch.device("cpu"))

        def add_state(self, name, default, dist_reduce_fx=None):
            self._defaults[name] = default.clone()
            setattr(self, name, default.clone())

        def reset(self):
            for k, v in self._defaults.items():
                setattr(self, k, v.clone())

    mg.stub("torchmetrics", Metric=Metric)
    mg.stub("py4cast.datasets", get_datasets=None)
    mg.stub("py4cast.datasets.base", DatasetInfo=object, NamedTensor=NT)
    mg.stub("py4cast.plots", plot_log_psd=None)
    sys.path.insert(0, mg.REF)
    metrics = importlib.import_module("py4cast.metrics")  # unmodified reference file

    g = torch.Generator().manual_seed(77)
    B, T, H, W, F = 2, 3, 12, 10, 5
    names = [f"f{i}" for i in range(F)]
    clim = torch.randn(F, generator=g) * 0.3

    class Info:
        shortnames = {"input_output": names[:3], "output": names[3:]}
        stats = mg.StatsLike({n: {"mean": clim[i]} for i, n in enumerate(names)})

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = metrics.MetricACC(Info())
    out = {"clim": clim.numpy()}
    for step in range(2):
        p = torch.randn(B, T, H, W, F, generator=g)
        t = torch.randn(B, T, H, W, F, generator=g)
        mask = (torch.rand(B, T, H, W, F, generator=g) > 0.1) if step == 1 else torch.ones(B, T, H, W, F, dtype=torch.bool)
        dims = ["batch", "timestep", "lat", "lon", "features"]
        m.update(NT(p, dims, names), NT(t, dims, names), mask)
        out[f"pred{step}"], out[f"target{step}"], out[f"mask{step}"] = p.numpy(), t.numpy(), mask.numpy()
        out[f"sum_acc{step}"] = m.sum_acc.clone().numpy()
    res = m.compute(prefix="val")
    out["compute_keys"] = np.array(sorted(res.keys()))
    out["compute_vals"] = np.array([float(res[k]) for k in sorted(res.keys())], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "next_acc.npz"), **out)
    for k in ("torchmetrics", "py4cast.datasets", "py4cast.datasets.base", "py4cast.plots", "py4cast.metrics", "py4cast"):
        sys.modules.pop(k, None)


def gen_unnormalize():
    losses, lightning = mg.install_stubs()
    src = inspect.getsource(lightning.AutoRegressiveLightning.predict_step).splitlines()
    start = next(i for i, l in enumerate(src) if "# Unnormalize data" in l)
    end = next(i for i in range(start + 1, len(src)) if src[i].strip() == "")
    snippet = textwrap.dedent("\n".join(src[start:end]))  # the reference's own lines, executed as they stand
    g = torch.Generator().manual_seed(5)
    B, T, H, W, F = 2, 2, 6, 7, 5
    names = [f"f{i}" for i in range(F)]
    std, mean = torch.rand(F, generator=g) * 3 + 0.1, torch.randn(F, generator=g) * 10

    class Self:
        stats = mg.StatsLike({n: {"mean": mean[i], "std": std[i]} for i, n in enumerate(names)})

    x = torch.randn(B, T, H, W, F, generator=g)
    preds = mg.NamedTensor(x.clone(), ["batch", "timestep", "lat", "lon", "features"], names)
    exec(snippet, {"torch": torch}, {"self": Self(), "preds": preds})
    np.savez_compressed(os.path.join(HERE, "next_unnormalize.npz"), x=x.numpy(), std=std.numpy(), mean=mean.numpy(),
                        out=preds.tensor.numpy())
    for k in [k for k in sys.modules if k.startswith("py4cast")]:
        sys.modules.pop(k, None)


def gen_pack():
    any_ = type("Any", (), {})
    mg.stub("gif", frame=lambda f: f)
    mg.stub("mfai"); mg.stub("mfai.pytorch"); mg.stub("mfai.pytorch.namedtensor", NamedTensor=NT)
    # a package stub whose __path__ is the real directory: `py4cast.datasets.base` is the reference's file, the package
    # __init__ (which imports every dataset backend) is not executed
    mg.stub("py4cast.datasets", __path__=[os.path.join(mg.REF, "py4cast", "datasets")])
    mg.stub("py4cast.datasets.access", DataAccessor=any_, Grid=any_, Period=any_, SamplePreprocSettings=any_, Stats=any_,
            Timestamps=any_, WeatherParam=any_, grid_static_features=None)
    mg.stub("py4cast.forcingutils", generate_toa_radiation_forcing=None, get_year_hour_forcing=None)
    mg.stub("py4cast.plots", DomainInfo=any_)
    mg.stub("py4cast.utils", RegisterFieldsMixin=object, merge_dicts=None)
    sys.path.insert(0, mg.REF)
    base = importlib.import_module("py4cast.datasets.base")  # unmodified reference file

    rng = np.random.default_rng(11)
    B, T, H, W, F = 2, 4, 6, 5, 7
    names = [f"p{i}" for i in range(F)]
    raw = (rng.standard_normal((F, B, T, H, W)) * rng.uniform(0.5, 30, (F, 1, 1, 1, 1)) + rng.uniform(-50, 300, (F, 1, 1, 1, 1))).astype(np.float32)
    mean = torch.tensor(rng.uniform(-50, 300, F), dtype=torch.float32)
    std = torch.tensor(rng.uniform(0.5, 30, F), dtype=torch.float32)

    class Accessor:
        def __init__(self, b):
            self.b = b

        def load_data_from_disk(self, ds, param, timestamps, member, fmt):
            return raw[param, self.b][..., None]  # (T,H,W,1), the layout titan/__init__.py returns

        def parameter_namer(self, param):
            return names[param]

    items = []
    for b in range(B):
        self_ = types.SimpleNamespace(accessor=Accessor(b), settings=types.SimpleNamespace(dataset_name="x", file_format="npy"),
                                      member=0, stats={n: {"mean": mean[i], "std": std[i]} for i, n in enumerate(names)})
        nts = [NT(base.Sample.get_param_tensor(self_, f, None, True), ["timestep", "lat", "lon", "features"], [names[f]])
               for f in range(F)]
        full = NT.concat(nts)
        items.append(base.Item(inputs=NT(full.tensor[:1], full.names, full.feature_names), forcing=NT(full.tensor[1:], full.names, full.feature_names),
                               outputs=NT(full.tensor[1:], full.names, full.feature_names), validity_times=[]))
    batch = base.collate_fn(items)
    np.savez_compressed(os.path.join(HERE, "next_pack.npz"), raw=raw, mean=mean.numpy(), std=std.numpy(),
                        inputs=batch.inputs.tensor.numpy(), outputs=batch.outputs.tensor.numpy())




def gen_stats():
    """compute_mean_std_min_max / compute_time_step_stats of the unmodified compute_dataset_stats.py on a fake dataset."""
    for k in [k for k in sys.modules if k.startswith("py4cast")]:
        sys.modules.pop(k, None)
    mg.stub("py4cast")
    sys.modules["py4cast"].__path__ = [os.path.join(mg.REF, "py4cast")]
    mg.stub("py4cast.datasets", __path__=[os.path.join(mg.REF, "py4cast", "datasets")])
    mg.stub("py4cast.datasets.base", DatasetABC=object)
    saved = {}
    mg.stub("py4cast.utils", torch_save=lambda obj, path: saved.__setitem__(str(path), obj))
    cds = importlib.import_module("py4cast.datasets.compute_dataset_stats")  # unmodified reference file

    g = torch.Generator().manual_seed(21)
    B, T, H, W = 2, 3, 6, 5
    names = {"inputs": ["a", "b", "c"], "outputs": ["a", "b", "c"], "forcing": ["f0", "f1"]}
    dims = ["batch", "timestep", "lat", "lon", "features"]

    def batch(nan):
        out = {}
        for kind, steps in (("inputs", 1), ("outputs", T), ("forcing", T)):
            t = torch.randn(B, steps, H, W, len(names[kind]), generator=g) * 3 + 1
            if nan and kind != "forcing":
                t[torch.rand(t.shape, generator=g) < 0.05] = float("nan")
            out[kind] = NT(t, dims, names[kind])
        return types.SimpleNamespace(**out)

    batches = [batch(False), batch(True), batch(True)]

    class DS:
        cache_dir = __import__("pathlib").Path("/nonexistent")

        def __init__(self, standardize):
            self.settings = types.SimpleNamespace(standardize=standardize)

        def torch_dataloader(self):
            return list(batches)

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = {}
        for kind in ("inputs", "outputs", "forcing"):
            st = cds.compute_mean_std_min_max(DS(False), kind)
            for n, d in st.items():
                for k, v in d.items():
                    out[f"{kind}__{n}__{k}"] = np.float32(v)
        cds.compute_time_step_stats(DS(True))
    diff = saved["/nonexistent/diff_stats.pt"]
    for n, d in diff.items():
        for k, v in d.items():
            out[f"diff__{n}__{k}"] = np.float32(v)
    for i, b in enumerate(batches):
        for kind in ("inputs", "outputs", "forcing"):
            out[f"batch{i}__{kind}"] = getattr(b, kind).tensor.numpy()
    np.savez_compressed(os.path.join(HERE, "next_stats.npz"), **out)




if __name__ == "__main__":
    gen_acc()
    gen_unnormalize()
    gen_pack()
    gen_stats()
    print("written:", sorted(f for f in os.listdir(HERE) if f.startswith("next_")))
