"""
Validation metric next to the losses: the anomaly correlation coefficient of
``py4cast/metrics.py:355-455`` with the spatial reductions done by one HIP kernel (``p4c_acc_sums``) over
prediction and target, instead of five full-tensor torch passes.  Same ``update`` / ``compute`` / ``reset``
contract as the reference's torchmetrics ``Metric`` (state ``sum_acc`` (T,F) and ``step_count``, both summed
across ranks by the caller as ``dist_reduce_fx="sum"`` prescribes).
"""

import warnings

import torch

from . import ops
from .namedtensor import NamedTensor


class MetricACC:
    def __init__(self, dataset_info, device=None):
        warnings.warn(
            "ACC supposes access to climate normals; they are one scalar per field here (metrics.py:362-369)."
        )
        names = dataset_info.shortnames["input_output"] + dataset_info.shortnames["output"]
        self.climate_means = dataset_info.stats.to_list("mean", names)
        if device is not None:
            self.climate_means = self.climate_means.to(device)
        self.reset()

    def reset(self):
        self.sum_acc = torch.tensor(0.0)
        self.step_count = 0.0
        self.feature_names, self.pred_steps = None, None

    def update(self, preds: NamedTensor, target: NamedTensor, mask: torch.Tensor, *args):
        """(B,T,*S,F) prediction / target and a 0/1 mask of the same shape (metrics.py:387-433)."""
        if preds.tensor.shape != target.tensor.shape:
            raise ValueError("preds and target must have the same shape")
        if self.step_count == 0:
            self.climate_means = self.climate_means.to(preds.tensor.device)
            self.feature_names = preds.feature_names
            self.pred_steps = preds.tensor.shape[1]
        from .losses import _mask_spec   # the lazy markers of get_mask_on_nan are read in place (no mask / clean target built)

        spec, tgt = _mask_spec(mask, target)
        sums = ops.acc_sums(preds.tensor, tgt, spec, self.climate_means)
        res = torch.mean(sums[0] / torch.sqrt(sums[1] * sums[2]), dim=0)  # tiny (B,T,F) tail, metrics.py:425
        if not self.sum_acc.ndim:
            self.sum_acc = torch.zeros(self.pred_steps, preds.tensor.shape[-1], device=preds.tensor.device)
        self.sum_acc += res
        self.step_count += 1

    def compute(self, prefix: str = "val") -> dict:
        mean_acc = self.sum_acc / self.step_count
        out = {
            f"{prefix}_acc/{name}_step{j}": mean_acc[j, i]
            for i, name in enumerate(self.feature_names)
            for j in range(self.pred_steps)
        }
        self.reset()
        return out
