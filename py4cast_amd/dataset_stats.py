"""
Dataset statistics passes, mirror of ``py4cast/datasets/compute_dataset_stats.py``: same loops over the
dataset's dataloader, same results (including the reference's quirk of taking the per-batch min / max from the FIRST
sample of each batch, ``.values[0]`` at :48,52), with the per-batch reductions -- ``nanmean`` / ``nansum`` / NaN-aware
min / max over (time, lat, lon), or over time-step differences -- done by ONE pass of ``p4c_nan_moments`` per batch
instead of ~10 full-tensor torch passes.
"""

import warnings
from typing import Dict

import torch

from . import _lib as L


def nan_moments(x: torch.Tensor, x_next: torch.Tensor = None) -> torch.Tensor:
    """x: (B, ..., F) (a batch-strided view is fine as long as each sample is contiguous) -> (5,B,F):
    sum, sum of squares, non-NaN count, min, max of x (or of x_next - x) over everything between batch and features."""
    L.require_cuda(x)
    B, F = x.shape[0], x.shape[-1]
    rows = x[0].numel() // F
    if not x[0].is_contiguous() or x.dtype != torch.float32:
        x = x.float().contiguous()
    if x_next is not None and (not x_next[0].is_contiguous() or x_next.dtype != torch.float32 or x_next.stride(0) != x.stride(0)):
        x_next, x = x_next.float().contiguous(), x.float().contiguous()
    out = torch.empty(5, B, F, dtype=torch.float32, device=x.device)
    ws = torch.empty(5 * L.lib().p4c_loss_workspace_bytes(B, 1, rows, F) // 4, dtype=torch.float32, device=x.device)
    L.call("p4c_nan_moments", L.ptr(x), L.ptr(x_next), x.stride(0), L.ptr(out), L.ptr(ws), B, rows, F, L.stream(x.device))
    return out


def _batch_sums(m: torch.Tensor):
    """per-batch increments of compute_dataset_stats.py:45-46 / :107-108 from the (5,B,F) moments"""
    mean_bf = m[0] / m[2]            # nanmean over X (NaN where a (b,f) has no valid value, dropped by nansum below)
    sq_bf = m[1] / m[2]
    return torch.nansum(mean_bf, dim=0), torch.nansum(sq_bf, dim=0)


def compute_mean_std_min_max(dataset, type_tensor: str, device=None) -> Dict[str, Dict[str, torch.Tensor]]:
    """compute_dataset_stats.py:11-68."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    random_batch = next(iter(dataset.torch_dataloader()))
    named_tensor = getattr(random_batch, type_tensor)
    n_features = len(named_tensor.feature_names)
    sum_means = torch.zeros(n_features, device=device)
    sum_squares = torch.zeros(n_features, device=device)
    first = named_tensor.tensor.to(device)
    m0 = nan_moments(first.reshape(1, -1, n_features))      # the flattened random batch (:22-24)
    best_min, best_max = m0[3, 0], m0[4, 0]
    if bool((m0[2, 0] < first.numel() // n_features).any()):
        warnings.warn("Your dataset contain NaN values, statistics will be calculated ignoring the NaN.")
    counter = 0
    if dataset.settings.standardize:
        raise ValueError("Your dataset should not be standardized.")
    for batch in dataset.torch_dataloader():
        tensor = getattr(batch, type_tensor).tensor.to(device)
        counter += tensor.shape[0]
        m = nan_moments(tensor)
        s1, s2 = _batch_sums(m)
        sum_means += s1
        sum_squares += s2
        best_min = torch.minimum(best_min, m[3, 0])  # first sample of the batch only, as the reference (:48-49)
        best_max = torch.maximum(best_max, m[4, 0])
    mean = sum_means / counter
    std = torch.sqrt(sum_squares / counter - mean**2)
    return {name: {"mean": mean[i].cpu(), "std": std[i].cpu(), "min": best_min[i].cpu(), "max": best_max[i].cpu()}
            for i, name in enumerate(named_tensor.feature_names)}


def compute_parameters_stats(dataset, device=None) -> Dict[str, Dict[str, torch.Tensor]]:
    """compute_dataset_stats.py:71-85 (returns the dict the reference saves as parameters_stats.pt)."""
    all_stats = {}
    for type_tensor in ["inputs", "outputs", "forcing"]:
        for feature, stats in compute_mean_std_min_max(dataset, type_tensor, device).items():
            if feature not in all_stats:
                all_stats[feature] = stats
    return all_stats


def compute_time_step_stats(dataset, device=None) -> Dict[str, Dict[str, torch.Tensor]]:
    """compute_dataset_stats.py:88-127 (returns the dict the reference saves as diff_stats.pt)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    if not dataset.settings.standardize:
        raise ValueError("Your dataset should be standardized.")
    sum_means = sum_squares = None
    counter = 0
    batch = None
    for batch in dataset.torch_dataloader():
        in_out = torch.cat([batch.inputs.tensor, batch.outputs.tensor], dim=1).to(device).float().contiguous()
        counter += in_out.shape[0]
        m = nan_moments(in_out[:, :-1], in_out[:, 1:])       # diff = in_out[:, 1:] - in_out[:, :-1]
        s1, s2 = _batch_sums(m)
        sum_means = s1 if sum_means is None else sum_means + s1
        sum_squares = s2 if sum_squares is None else sum_squares + s2
    diff_mean = sum_means / counter
    diff_std = torch.sqrt(sum_squares / counter - diff_mean**2)
    store = {name: {"mean": diff_mean[i].cpu(), "std": diff_std[i].cpu()} for i, name in enumerate(batch.inputs.feature_names)}
    for name in batch.forcing.feature_names:
        store[name] = {"mean": torch.tensor(0), "std": torch.tensor(1)}
    return store
