"""
Output staging for the inference path: what happens between the rollout and the GRIB / GIF writers.

The reference (py4cast/lightning.py:1162-1188, py4cast/io/outputs.py:116-241) un-normalises the prediction feature by feature
on the device, then the writers pull ONE (lat, lon) plane per time step and feature with ``tensor[:, :, idx].cpu().numpy()``: a
strided gather plus a synchronous device-to-host copy per plane (3 steps x 60 features = 180 blocking copies per sample).

Here the un-normalisation kernel writes feature-major planes (``p4c_unnormalize_planes``: (B,T,*S,F) -> (B,T,F,*S), bit-exact
with the reference's two rounded steps) and the planes travel to PINNED host buffers with ONE asynchronous copy per batch on a
copy stream.  Two buffers rotate, so the next batch's rollout, the copy of this batch and the host-side writers (CPU-bound:
epygram / matplotlib) of the previous one overlap.  The writers themselves (GRIB, GIF) are the reference's host code and stay
out of the hot-path scope; they receive numpy views of the pinned planes -- no further copies.
"""

from dataclasses import dataclass
from typing import Callable, List, Optional

import numpy as np
import torch

from . import ops
from .namedtensor import NamedTensor


@dataclass
class StagedPrediction:
    """A batch of un-normalised predictions on the host: ``planes[b, t, f]`` is the contiguous (*spatial) field the writers take."""

    planes: np.ndarray            # (B, T, F, *spatial) float32 view of a pinned buffer -- valid until the slot is reused
    feature_names: List[str]
    names: List[str]              # dimension names of the un-staged tensor (batch, timestep, *spatial, features)
    slot: int

    def plane(self, b: int, t: int, feature: str) -> np.ndarray:
        return self.planes[b, t, self.feature_names.index(feature)]


class OutputStager:
    """``submit`` enqueues kernel + copy and returns at once; ``wait`` blocks on that batch's copy only."""

    def __init__(self, device: torch.device, slots: int = 2):
        if device.type != "cuda":
            raise RuntimeError("OutputStager stages device predictions: it needs the GPU (no CPU fallback)")
        self.device, self.slots = device, slots
        self.copy_stream = torch.cuda.Stream(device=device)
        self._host: List[Optional[torch.Tensor]] = [None] * slots
        self._dev: List[Optional[torch.Tensor]] = [None] * slots
        self._done: List[Optional[torch.cuda.Event]] = [None] * slots
        self._meta = [None] * slots
        self._next = 0

    def _buffers(self, slot: int, shape):
        n = int(np.prod(shape))
        if self._host[slot] is None or self._host[slot].numel() < n:
            self._host[slot] = torch.empty(n, dtype=torch.float32, pin_memory=True)
            self._dev[slot] = torch.empty(n, dtype=torch.float32, device=self.device)
        return self._host[slot][:n].view(shape), self._dev[slot][:n].view(shape)

    def submit(self, preds: NamedTensor, std: torch.Tensor, mean: torch.Tensor) -> int:
        """preds: normalised (B,T,*S,F) on the device.  Returns the slot to pass to ``wait``."""
        slot = self._next
        self._next = (self._next + 1) % self.slots
        if self._done[slot] is not None:
            self._done[slot].synchronize()   # the slot's previous batch must have left the device before its buffers are reused
        t = preds.tensor
        shape = (t.shape[0], t.shape[1], t.shape[-1]) + tuple(t.shape[2:-1])
        host, dev = self._buffers(slot, shape)
        ops.unnormalize_planes(t, std, mean, out=dev)                 # compute stream
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(ready)
            host.copy_(dev, non_blocking=True)                        # one DMA for the whole batch, pinned destination
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        self._done[slot] = done
        self._meta[slot] = (host, list(preds.feature_names), list(preds.names))
        return slot

    def wait(self, slot: int) -> StagedPrediction:
        self._done[slot].synchronize()
        host, feature_names, names = self._meta[slot]
        return StagedPrediction(host.numpy(), feature_names, names, slot)

    def run(self, batches, predict: Callable, std: torch.Tensor, mean: torch.Tensor, write: Callable[[StagedPrediction, int], None]):
        """The inference loop with the overlap spelled out: batch i's writers run while batch i+1 is on the device."""
        pending = None
        for i, batch in enumerate(batches):
            slot = self.submit(predict(batch, i), std, mean)
            if pending is not None:
                write(self.wait(pending[0]), pending[1])
            pending = (slot, i)
        if pending is not None:
            write(self.wait(pending[0]), pending[1])
