"""Shared builders for the tests: synthetic DatasetInfo / batches shaped like the reference's."""

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from py4cast_amd.base import DatasetInfo, ItemBatch, ModelABC, ModelType, Statics, Stats
from py4cast_amd.namedtensor import NamedTensor

GRID_DIMS = ["batch", "timestep", "lat", "lon", "features"]


def feature_names(F):
    return [f"f{i}" for i in range(F)]


def make_dataset_info(case: dict, Ff: int) -> DatasetInfo:
    """case: dict of tensors as produced by tests/golden/make_golden.py::make_case (or alike)."""
    F = case["diff_std"].shape[0]
    names = feature_names(F)
    st = case["statics"].clone()  # (H,W,Fs): x, y, geopotential, border_mask (access.py:297-305)
    st[..., 3:4] = case["border_mask"]
    grid_statics = NamedTensor(st, ["lat", "lon", "features"], ["x", "y", "geopotential", "border_mask"][: st.shape[-1]])
    statics = Statics(grid_statics, tuple(st.shape[:2]))
    stats = Stats({n: {"std": case["std"][i], "mean": torch.tensor(0.0)} for i, n in enumerate(names)})
    diff_stats = Stats({n: {"std": case["diff_std"][i], "mean": case["diff_mean"][i]} for i, n in enumerate(names)})
    return DatasetInfo(
        name="synthetic",
        statics=statics,
        stats=stats,
        diff_stats=diff_stats,
        state_weights={n: float(case["state_weight"][i]) for i, n in enumerate(names)},
        shortnames={"input_output": names},
        weather_dim=F,
        forcing_dim=Ff,
    )


def make_batch(case: dict, device) -> ItemBatch:
    F, Ff = case["inputs"].shape[-1], case["forcing"].shape[-1]
    return ItemBatch(
        NamedTensor(case["inputs"].clone().to(device), GRID_DIMS, feature_names(F)),
        NamedTensor(case["forcing"].clone().to(device), GRID_DIMS, [f"g{i}" for i in range(Ff)]),
        NamedTensor(case["outputs"].clone().to(device), GRID_DIMS, feature_names(F)),
    )


@dataclass
class TinySettings:
    name: str = "tiny"


class TinyConvModel(ModelABC, torch.nn.Module):
    """The golden fixtures' model: 3x3 conv + tanh, NCHW (features_second), plain torch ops."""

    settings_kls = TinySettings
    onnx_supported = False
    supported_num_spatial_dims = (2,)
    num_spatial_dims = 2
    features_last = False
    model_type = ModelType.CONVOLUTIONAL
    register = True

    def __init__(self, in_channels, out_channels, input_shape, settings=None, *args, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.input_shape = in_channels, out_channels, input_shape
        self._settings = settings
        self.w = torch.nn.Parameter(torch.zeros(out_channels, in_channels, 3, 3))
        self.b = torch.nn.Parameter(torch.zeros(out_channels))
        self.check_required_attributes()

    @property
    def settings(self):
        return self._settings

    def forward(self, x):
        return torch.tanh(torch.nn.functional.conv2d(x, self.w, self.b, padding=1))


class TinyLinearModel(ModelABC, torch.nn.Module):
    """Graph-layout golden model: per-node linear + tanh on (B,N,C)."""

    settings_kls = TinySettings
    onnx_supported = False
    supported_num_spatial_dims = (1,)
    num_spatial_dims = 1
    features_last = True
    model_type = ModelType.GRAPH
    register = True

    def __init__(self, in_channels, out_channels, input_shape, settings=None, *args, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.input_shape = in_channels, out_channels, input_shape
        self._settings = settings
        self.w = torch.nn.Parameter(torch.zeros(in_channels, out_channels))
        self.b = torch.nn.Parameter(torch.zeros(out_channels))
        self.check_required_attributes()

    @property
    def settings(self):
        return self._settings

    def forward(self, x):
        return torch.tanh(x @ self.w + self.b)


def register_test_models():
    from py4cast_amd.models import registry

    registry.setdefault("TinyConvModel", TinyConvModel)
    registry.setdefault("TinyLinearModel", TinyLinearModel)


def synthetic_case(seed=0, B=2, T=3, T_in=1, H=32, W=32, F=60, Ff=5, Fs=4, border=0, nan=False):
    """Seeded synthetic tensors following SURVEY.md section 8(d)."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    ru = lambda *s: torch.rand(*s, generator=g)
    forcing = ru(B, T, H, W, Ff)
    forcing[..., -1] *= 1366.0
    case = dict(
        inputs=rn(B, T_in, H, W, F).clamp(-3, 3),
        forcing=forcing,
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
    case["border_mask"] = bm
    case["statics"][..., 3:4] = bm
    if nan:
        case["inputs"][0, 0, 3, 4, 1] = float("nan")
        case["forcing"][-1, -1, 5, 6, 2] = float("nan")
        case["outputs"][:, :, 7, 8, :] = float("nan")
        case["outputs"][0, 1, 2, 2, 0] = float("nan")
    return case
