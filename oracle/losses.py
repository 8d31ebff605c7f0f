"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of ``/root/reference/py4cast/losses.py`` on raw tensors.

Shapes (grid layout): prediction/target/mask (B,T,H,W,F); interior_mask (H,W,1).
``kind`` is "mse" (torch.nn.MSELoss(reduction="none")) or "l1" (L1Loss).

Graph layout (B,T,N,F): the reference raises IndexError (losses.py:156,197 hard-code
``dim=(0, 1, 4)``).  This restatement defines it as the grid-layout value of the same
tensors (union over batch, time and features; reduce over the single spatial dim).
"""

from typing import List, Sequence, Tuple

import torch


def _elementwise(pred: torch.Tensor, tgt: torch.Tensor, kind: str) -> torch.Tensor:
    if kind == "mse":
        return (pred - tgt) ** 2
    if kind == "l1":
        return (pred - tgt).abs()
    raise NameError(f"Loss: {kind} is not defined")


def weighted_loss_weights(state_weight: torch.Tensor, diff_std: torch.Tensor, kind: str) -> torch.Tensor:
    """losses.py:110-124: w_f = state_weight_f / diff_std_f ** (2 if MSE else 1)."""
    exponent = 2.0 if kind == "mse" else 1.0
    return state_weight / (diff_std**exponent)


def _denominator(mask: torch.Tensor, num_interior: float) -> torch.Tensor:
    # losses.py:156,167 / 197,203: union over batch, time, features -> spatial map;
    # note the bug-compatible count of *all* fully-masked pixels (border ones included).
    feat = mask.dim() - 1
    union_mask = torch.any(mask != 0, dim=(0, 1, feat)) if mask.dtype != torch.bool else torch.any(mask, dim=(0, 1, feat))
    return num_interior - (~union_mask).sum()


def weighted_loss(
    prediction: torch.Tensor,
    target: torch.Tensor,
    mask: torch.Tensor,
    weights: torch.Tensor,
    interior_mask: torch.Tensor,
    kind: str = "mse",
    reduce_spatial_dim: bool = True,
) -> torch.Tensor:
    """losses.py:130-169 (WeightedLoss.forward).  Returns (B,T) or (B,T,H,W)."""
    e = _elementwise(prediction * mask, target * mask, kind)  # :144
    wl = torch.sum(e * weights, dim=-1)  # :150
    if not reduce_spatial_dim:
        return wl
    interior_s = interior_mask.squeeze(-1)  # :65-71
    num_interior = torch.sum(interior_mask).item()  # :72
    spatial = tuple(range(2, prediction.dim() - 1))
    return torch.sum(wl * interior_s, dim=spatial) / _denominator(mask, num_interior)  # :164-167


def scaled_loss(
    prediction: torch.Tensor,
    target: torch.Tensor,
    mask: torch.Tensor,
    std: torch.Tensor,
    interior_mask: torch.Tensor,
    kind: str = "mse",
) -> torch.Tensor:
    """losses.py:186-210 (ScaledLoss.forward).  Returns (B,T,F)."""
    e = _elementwise(prediction * mask, target * mask, kind)  # :195
    num_interior = torch.sum(interior_mask).item()
    spatial = tuple(range(2, prediction.dim() - 1))
    mean_loss = torch.sum(e * interior_mask, dim=spatial) / _denominator(mask, num_interior)  # :200-203
    if kind == "mse":
        mean_loss = torch.sqrt(mean_loss)  # :205-206
    return mean_loss * std  # :208-210


def combined_loss(
    prediction: torch.Tensor,
    target: torch.Tensor,
    mask: torch.Tensor,
    members: Sequence[Tuple[str, float, dict]],
    reduce_spatial_dim: bool = True,
) -> torch.Tensor:
    """
    losses.py:286-307 (CombinedLoss.forward).  ``members`` = [(class_name, weight, kwargs)]
    with kwargs the arguments of weighted_loss / scaled_loss besides (prediction,target,mask).
    """
    shape = prediction.shape[:2] if reduce_spatial_dim else prediction.shape[:-1]
    total = torch.zeros(shape, dtype=prediction.dtype, device=prediction.device)
    for name, weight, kw in members:
        if name == "WeightedLoss":
            total += weight * weighted_loss(prediction, target, mask, reduce_spatial_dim=reduce_spatial_dim, **kw)
        elif name == "ScaledLoss":
            if not reduce_spatial_dim:
                raise TypeError("ScaledLoss.forward() got an unexpected keyword argument 'reduce_spatial_dim'")
            total += weight * scaled_loss(prediction, target, mask, **kw)  # (B,T) += (B,T,F) raises, as in the reference
        else:
            raise KeyError(name)
    return total


def training_loss(
    prediction: torch.Tensor, target: torch.Tensor, mask_on_nan: bool, members, **kw
) -> torch.Tensor:
    """lightning.py:811-816: NaN mask on the target then mean over (B,T) of the combined loss."""
    from .rollout import get_mask_on_nan

    mask, target_masked = get_mask_on_nan(target, mask_on_nan)
    return torch.mean(combined_loss(prediction, target_masked, mask, members, **kw))
