"""
Input pipeline step right before the hot path: the reference standardises every parameter on the
CPU with numpy (``Sample.get_param_tensor``, datasets/base.py:431-453), concatenates the parameters along the feature
axis (``Sample.load`` :455-527) and stacks samples (``collate_fn`` :173-195).  Here the RAW parameter planes go to the
device as they come from disk (one plane per parameter) and ONE kernel standardises and packs them into the
features-last batch layout the rollout consumes.
"""

from typing import Dict, List, Sequence

import torch

from . import ops
from .base import ItemBatch
from .namedtensor import NamedTensor

DIMS = ["batch", "timestep", "lat", "lon", "features"]


def standardize_and_collate(raw: torch.Tensor, feature_names: Sequence[str], stats, standardize: bool = True) -> NamedTensor:
    """raw: (F, B, T, H, W) device tensor of un-normalised parameter planes -> NamedTensor (B,T,H,W,F) fp32."""
    F = raw.shape[0]
    if standardize:
        mean = stats.to_list("mean", list(feature_names)).to(raw.device)
        std = stats.to_list("std", list(feature_names)).to(raw.device)
    else:
        mean, std = torch.zeros(F, device=raw.device), torch.ones(F, device=raw.device)
    return NamedTensor(ops.pack_standardize(raw, mean, std), DIMS.copy(), list(feature_names))


def load_batch(raw_io: torch.Tensor, io_names: List[str], forcing: NamedTensor, stats, num_input_steps: int,
               standardize: bool = True) -> ItemBatch:
    """Device-side ``Sample.load`` + ``collate_fn`` for input_output parameters: the first ``num_input_steps`` time
    steps are the inputs, the rest the targets (base.py:493-505)."""
    full = standardize_and_collate(raw_io, io_names, stats, standardize)
    t = full.tensor
    inputs = NamedTensor(t[:, :num_input_steps], DIMS.copy(), list(io_names))
    outputs = NamedTensor(t[:, num_input_steps:], DIMS.copy(), list(io_names))
    return ItemBatch(inputs=inputs, forcing=forcing, outputs=outputs)
