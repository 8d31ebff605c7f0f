"""
NamedTensor: metadata wrapper (dim names + feature names) around one tensor.

py4cast takes this class from the third-party package ``mfai`` (py4cast/lightning.py:27,
py4cast/losses.py:12).  When ``mfai`` is importable we re-export its class untouched; the
local class below is only a fallback so that this package works stand-alone (mfai is absent
from the build image).  It implements the subset of the API the hot path uses
(lightning.py:528-535,542-545,567,599,613-615,638-656,667-675,722-727,764; losses.py:147,166,202).
"""

from typing import List, Sequence

import torch

try:  # pragma: no cover - mfai is not installed in the build image
    from mfai.pytorch.namedtensor import NamedTensor  # type: ignore  # noqa: F401

    HAVE_MFAI = True
except Exception:  # ImportError or transitive failures
    HAVE_MFAI = False

    class NamedTensor:
        SPATIAL_DIM_NAMES = ("lat", "lon", "ngrid")

        def __init__(
            self,
            tensor: torch.Tensor,
            names: Sequence[str],
            feature_names: Sequence[str],
            feature_dim_name: str = "features",
        ):
            if tensor.dim() != len(names):
                raise ValueError(f"Number of names ({len(names)}) != number of dims ({tensor.dim()})")
            if tensor.shape[list(names).index(feature_dim_name)] != len(feature_names):
                raise ValueError(
                    f"Number of feature names ({len(feature_names)}) does not match the feature dim "
                    f"({tensor.shape[list(names).index(feature_dim_name)]})"
                )
            self.tensor = tensor
            self.names = list(names)
            self.feature_names = list(feature_names)
            self.feature_dim_name = feature_dim_name
            self.feature_names_to_idx = {n: i for i, n in enumerate(self.feature_names)}

        # -- metadata -------------------------------------------------------------
        @property
        def device(self):
            return self.tensor.device

        @property
        def ndims(self) -> int:
            return len(self.names)

        @property
        def spatial_dim_idx(self) -> List[int]:
            return sorted(self.names.index(n) for n in set(self.SPATIAL_DIM_NAMES) & set(self.names))

        @property
        def num_spatial_dims(self) -> int:
            return len(self.spatial_dim_idx)

        def dim_size(self, name: str) -> int:
            try:
                return self.tensor.size(self.names.index(name))
            except ValueError as e:
                raise ValueError(f"Dimension {name} not found in {self.names}") from e

        def dim_index(self, name: str) -> int:
            return self.names.index(name)

        # -- selection ------------------------------------------------------------
        def select_tensor_dim(self, name: str, index: int) -> torch.Tensor:
            return self.tensor.select(self.names.index(name), index)

        def select_dim(self, name: str, index: int) -> "NamedTensor":
            return NamedTensor(
                self.select_tensor_dim(name, index),
                [n for n in self.names if n != name],
                self.feature_names,
                self.feature_dim_name,
            )

        def index_select_tensor_dim(self, name: str, indices) -> torch.Tensor:
            return self.tensor.index_select(
                self.names.index(name), torch.tensor(list(indices), dtype=torch.int64, device=self.device)
            )

        def __getitem__(self, feature_name: str) -> torch.Tensor:
            fdim = self.names.index(self.feature_dim_name)
            return self.tensor.select(fdim, self.feature_names_to_idx[feature_name]).unsqueeze(fdim)

        # -- construction ---------------------------------------------------------
        @staticmethod
        def new_like(tensor: torch.Tensor, other: "NamedTensor") -> "NamedTensor":
            return NamedTensor(tensor, other.names.copy(), other.feature_names.copy(), other.feature_dim_name)

        @staticmethod
        def expand_to_batch_like(tensor: torch.Tensor, other: "NamedTensor") -> "NamedTensor":
            return NamedTensor(tensor, ["batch"] + other.names, other.feature_names.copy(), other.feature_dim_name)

        def clone(self) -> "NamedTensor":
            return NamedTensor(self.tensor.clone(), self.names.copy(), self.feature_names.copy(), self.feature_dim_name)

        def flatten_(self, flatten_dim_name: str, start_dim: int, end_dim: int) -> None:
            self.tensor = torch.flatten(self.tensor, start_dim, end_dim)
            self.names = self.names[:start_dim] + [flatten_dim_name] + self.names[end_dim + 1 :]

        def unflatten_(self, dim: int, unflattened_size, unflatten_dim_name) -> None:
            self.tensor = self.tensor.unflatten(dim, unflattened_size)
            self.names = self.names[:dim] + list(unflatten_dim_name) + self.names[dim + 1 :]

        def to_(self, *args, **kwargs) -> None:
            self.tensor = self.tensor.to(*args, **kwargs)

        def pin_memory_(self) -> None:
            self.tensor = self.tensor.pin_memory()

        def type_(self, dtype) -> None:
            self.tensor = self.tensor.type(dtype)

        def __repr__(self) -> str:
            return f"NamedTensor(shape={tuple(self.tensor.shape)}, names={self.names}, features={self.feature_names})"
