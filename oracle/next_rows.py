"""
CPU restatement (TEST INFRASTRUCTURE, not a product path) of the rows next to the hot path:
validation anomaly correlation, inference un-normalisation, load-time standardisation + collation.
Pinned to ``tests/golden/next_*.npz`` (outputs of the unmodified reference, ``tests/golden/make_golden_next.py``)
by ``tests/test_oracle_golden.py``.
"""

import numpy as np
import torch


def acc_update(pred: torch.Tensor, target: torch.Tensor, mask: torch.Tensor, climate_means: torch.Tensor) -> torch.Tensor:
    """MetricACC.update (py4cast/metrics.py:387-433): (B,T,*S,F) -> the (T,F) increment of ``sum_acc``."""
    spatial = tuple(range(2, pred.dim() - 1))
    dp, dt = pred - climate_means, target - climate_means
    num = (dp * dt * mask).mean(dim=spatial)                                   # :414-418
    den = ((dp * mask) ** 2).mean(dim=spatial) * ((dt * mask) ** 2).mean(dim=spatial)  # :419-423
    return torch.mean(num / torch.sqrt(den), dim=0)                            # :425


def acc_compute(sum_acc: torch.Tensor, step_count: float, feature_names, prefix: str = "val") -> dict:
    """MetricACC.compute (metrics.py:440-455)."""
    mean_acc = sum_acc / step_count
    return {f"{prefix}_acc/{n}_step{j}": mean_acc[j, i] for i, n in enumerate(feature_names) for j in range(mean_acc.shape[0])}


def unnormalize(x: torch.Tensor, std: torch.Tensor, mean: torch.Tensor) -> torch.Tensor:
    """predict_step's per-feature ``*= std`` then ``+= mean`` (py4cast/lightning.py:1162-1169): two rounded steps."""
    out = x.clone()
    out *= std
    out += mean
    return out


def standardize_pack(raw: np.ndarray, mean: np.ndarray, std: np.ndarray) -> np.ndarray:
    """Sample.get_param_tensor standardisation (datasets/base.py:448-452: ``(arr - means) / std`` in the array's
    dtype) + NamedTensor.concat along features + collate_fn's stack and fp32 cast (:173-195).
    raw: (F, B, T, H, W) planes -> (B, T, H, W, F)."""
    planes = [(raw[f] - np.asarray(mean[f])) / np.asarray(std[f]) for f in range(raw.shape[0])]
    return np.stack(planes, axis=-1).astype(np.float32)


def nan_moments(x: torch.Tensor, x_next: torch.Tensor = None) -> torch.Tensor:
    """The per-(sample, feature) reductions compute_dataset_stats.py is built from (:45-52, :105-108):
    (5,B,F) = nansum, nansum of squares, non-NaN count, min / max with NaN ignored, over all dims between batch and
    features, of x or of x_next - x."""
    v = (x_next - x if x_next is not None else x).double()
    B, F = v.shape[0], v.shape[-1]
    v = v.reshape(B, -1, F)
    ok = ~torch.isnan(v)
    z = torch.where(ok, v, torch.zeros_like(v))
    mn = torch.min(torch.nan_to_num(v, nan=float("inf")), dim=1).values
    mx = torch.max(torch.nan_to_num(v, nan=float("-inf")), dim=1).values
    return torch.stack([z.sum(1), (z * z).sum(1), ok.sum(1).double(), mn, mx]).float()
