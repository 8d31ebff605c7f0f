"""
Host-side types of the hot path's boundary.

* ``ModelABC`` / ``ModelType``: py4cast imports them from ``mfai.pytorch.models.base``
  (py4cast/models.py:11, py4cast/lightning.py:16).  Real mfai classes are used when
  importable; the fallbacks keep the same attribute contract
  (doc/add_features_contribute.md:19-30 of the reference).
* ``ItemBatch``: batch container (py4cast/datasets/base.py:147-170).
* ``Stats`` / ``DatasetInfo`` / ``Statics``: the fields of the reference's dataclasses
  (datasets/access.py:355-390, datasets/base.py:198-230) that the rollout and the losses
  read.  Any object with the same attributes (e.g. the reference's own) works.
"""

from dataclasses import dataclass, field
from enum import Enum
from typing import Dict, List, Optional, Tuple

import torch

from .namedtensor import NamedTensor

try:  # pragma: no cover - mfai absent in the build image
    from mfai.pytorch.models.base import ModelABC, ModelType  # type: ignore  # noqa: F401
except Exception:

    class ModelType(Enum):
        GRAPH = 1
        CONVOLUTIONAL = 2
        VISION_TRANSFORMER = 3

    class ModelABC:
        """Attribute contract of a py4cast model plugin (see py4cast_plugin_example.py:19-57)."""

        register: bool = False
        # concrete classes define: settings_kls, onnx_supported, supported_num_spatial_dims,
        # num_spatial_dims, features_last, model_type, settings (property)

        @property
        def features_second(self) -> bool:
            return not self.features_last

        def check_required_attributes(self) -> None:
            for attr in ("in_channels", "out_channels", "input_shape"):
                if not hasattr(self, attr):
                    raise AttributeError(f"Missing required attribute : {attr}")


def features_last_to_second(x: torch.Tensor) -> torch.Tensor:
    """mfai.pytorch.models.utils.features_last_to_second (used at lightning.py:592)."""
    return x.movedim(-1, 1)


def features_second_to_last(x: torch.Tensor) -> torch.Tensor:
    return x.movedim(1, -1)


def expand_to_batch(x: torch.Tensor, batch_size: int) -> torch.Tensor:
    """mfai.pytorch.models.utils.expand_to_batch (used at lightning.py:298)."""
    return x.unsqueeze(0).expand(batch_size, *x.shape)


@dataclass
class ItemBatch:
    """inputs/outputs/forcing: NamedTensor (batch, timestep, lat, lon, features) -- base.py:147-170."""

    inputs: NamedTensor
    forcing: NamedTensor
    outputs: Optional[NamedTensor]

    @property
    def batch_size(self) -> int:
        return (self.outputs if self.outputs is not None else self.inputs).dim_size("batch")

    @property
    def num_input_steps(self) -> int:
        return self.inputs.dim_size("timestep")

    @property
    def num_pred_steps(self) -> int:
        return (self.outputs if self.outputs is not None else self.forcing).dim_size("timestep")


class Stats:
    """name -> {"mean","std","min","max": 0-d tensor}; ``to_list`` as access.py:368-390."""

    def __init__(self, stats: Dict[str, Dict[str, torch.Tensor]]):
        self.stats = stats

    def items(self):
        return self.stats.items()

    def __getitem__(self, name: str):
        return self.stats[name]

    def to_list(self, stat_name: str, shortnames: List[str], dtype: torch.dtype = torch.float32):
        if len(shortnames) > 0:
            return torch.stack([torch.as_tensor(self[n][stat_name]) for n in shortnames], dim=0).type(dtype)
        return []


@dataclass
class Statics:
    """grid_statics: NamedTensor (lat, lon, features) with a "border_mask" feature -- base.py:198-230."""

    grid_statics: NamedTensor
    grid_shape: Tuple[int, int]
    border_mask: torch.Tensor = field(init=False)
    interior_mask: torch.Tensor = field(init=False)

    def __post_init__(self):
        self.border_mask = self.grid_statics["border_mask"]
        self.interior_mask = 1.0 - self.border_mask

    @property
    def meshgrid(self) -> torch.Tensor:
        return torch.cat([self.grid_statics["x"], self.grid_statics["y"]], dim=-1).permute(2, 0, 1)

    def register_buffers(self, lm: torch.nn.Module, persistent: bool = False) -> None:
        # RegisterFieldsMixin.register_buffers (utils.py:74-89)
        for name in ("border_mask", "interior_mask"):
            lm.register_buffer(name, getattr(self, name), persistent=persistent)


@dataclass
class DatasetInfo:
    """The DatasetInfo fields the hot path reads (lightning.py:232-263, losses.py:121-124,182)."""

    name: str
    statics: Statics
    stats: Stats
    diff_stats: Stats
    state_weights: Dict[str, float]
    shortnames: Dict[str, List[str]]
    weather_dim: int
    forcing_dim: int
    pred_step: float = 1.0
    domain_info: object = None
    units: Optional[Dict[str, str]] = None

    def summary(self):
        print(f"Dataset {self.name}: weather_dim={self.weather_dim} forcing_dim={self.forcing_dim}")
