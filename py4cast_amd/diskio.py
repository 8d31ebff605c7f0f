"""
On-disk formats either side of the hot path, host side.

* Titan's training layout: one 2-D ``.npy`` plane per (date, parameter) --
  ``<dataset>/data/<date %Y-%m-%d_%Hh%M>/<name>_<level><hpa|m>.npy`` (datasets/titan/__init__.py:93-109,168-176), read by
  ``np.load`` per plane (:111-129), stacked per parameter and standardised on the CPU (datasets/base.py:431-453).
  Here the planes of a whole batch are read straight into ONE pinned host buffer (the ``.npy`` payload is ``readinto`` its slot:
  no intermediate arrays), copied to the device in one asynchronous transfer and standardised + packed into the features-last
  batch by one kernel (``datapipe.standardize_and_collate`` -> ``p4c_pack_standardize``).
* Statistics files: ``parameters_stats.pt`` / ``diff_stats.pt`` = ``torch.save`` of ``{name: {"mean","std","min","max": 0-d
  tensor}}`` (datasets/compute_dataset_stats.py:71-127, read by ``Stats`` at datasets/access.py:355-390, written by the dummy
  dataset at datasets/dummy.py:24-42).
"""

import datetime as dt
import os
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from numpy.lib import format as npy_format

from .base import Stats

TITAN_FORMATSTR = "%Y-%m-%d_%Hh%M"   # datasets/titan/settings.py:8


# ------------------------------------------------------------------------------------------ statistics files
def save_stats(stats: Dict[str, Dict[str, torch.Tensor]], fname) -> None:
    """The reference's layout: dict of dict of 0-d tensors, ``torch.save`` (compute_dataset_stats.py:84-85,126-127)."""
    out = {name: {k: torch.as_tensor(v).detach().cpu().reshape(()) for k, v in d.items()} for name, d in stats.items()}
    torch.save(out, fname)


def load_stats(fname) -> Stats:
    """``Stats(fname)`` of the reference (access.py:355-362)."""
    return Stats(torch.load(fname, "cpu", weights_only=True))


def write_dataset_stats(dataset, cache_dir, device=None) -> Tuple[Path, Path]:
    """compute_dataset_stats.py end to end on the device kernels: ``parameters_stats.pt`` then ``diff_stats.pt`` in ``cache_dir``."""
    from . import dataset_stats as ds

    cache_dir = Path(cache_dir)
    os.makedirs(cache_dir, exist_ok=True)
    p1, p2 = cache_dir / "parameters_stats.pt", cache_dir / "diff_stats.pt"
    save_stats(ds.compute_parameters_stats(dataset, device), p1)
    save_stats(ds.compute_time_step_stats(dataset, device), p2)
    return p1, p2


# ------------------------------------------------------------------------------------------ .npy planes
def titan_plane_path(dataset_path, name: str, level: int, level_type: str, date: dt.datetime) -> Path:
    """datasets/titan/__init__.py:93-109 (npy branch) with parameter_namer (:168-176)."""
    suffix = "m" if level_type in ("surface", "heightAboveGround") else "hpa"
    return Path(dataset_path) / "data" / date.strftime(TITAN_FORMATSTR) / f"{name}_{level}{suffix}.npy"


def _read_header(f):
    version = npy_format.read_magic(f)
    if version == (1, 0):
        shape, fortran, dtype = npy_format.read_array_header_1_0(f)
    elif version == (2, 0):
        shape, fortran, dtype = npy_format.read_array_header_2_0(f)
    else:
        raise ValueError(f"unsupported .npy version {version}")
    return shape, fortran, dtype


class NpyPlaneReader:
    """Reads (F, B, T) planes of H x W values into one pinned host buffer and returns them on the device as the
    (F, B, T, H, W) fp32 tensor ``datapipe.standardize_and_collate`` consumes."""

    def __init__(self, shape: Tuple[int, int], n_features: int, batch: int, steps: int, device=None, pin: Optional[bool] = None):
        self.shape = tuple(shape)
        self.dims = (n_features, batch, steps)
        self.device = device
        pin = torch.cuda.is_available() if pin is None else pin
        self.host = torch.empty(n_features, batch, steps, *self.shape, dtype=torch.float32, pin_memory=pin)
        self._np = self.host.numpy()     # same memory
        self.copy_stream = torch.cuda.Stream(device) if (device is not None and torch.device(device).type == "cuda") else None

    def _read_plane(self, path, dst: np.ndarray) -> None:
        with open(path, "rb") as f:
            shape, fortran, dtype = _read_header(f)
            if tuple(shape) != self.shape:
                raise ValueError(f"{path}: plane shape {tuple(shape)} != {self.shape}")
            if fortran:
                raise ValueError(f"{path}: Fortran-ordered planes are not supported")
            if dtype == np.dtype("<f4"):
                n = f.readinto(memoryview(dst).cast("B"))    # payload straight into the pinned slot
                if n != dst.nbytes:
                    raise ValueError(f"{path}: truncated file ({n} of {dst.nbytes} bytes)")
            else:                                            # other dtypes: numpy converts (np.load + astype in the reference's terms)
                dst[...] = np.fromfile(f, dtype=dtype, count=int(np.prod(shape))).reshape(shape).astype(np.float32)

    def read(self, paths: Sequence[Sequence[Sequence]]) -> torch.Tensor:
        """paths[f][b][t] -> (F, B, T, H, W) tensor on ``device`` (or the host buffer itself when device is None)."""
        F, B, T = self.dims
        if len(paths) != F or any(len(pb) != B or any(len(pt) != T for pt in pb) for pb in paths):
            raise ValueError(f"expected paths[{F}][{B}][{T}]")
        for fi in range(F):
            for b in range(B):
                for t in range(T):
                    self._read_plane(paths[fi][b][t], self._np[fi, b, t])
        if self.device is None:
            return self.host
        if self.copy_stream is None:
            return self.host.to(self.device)
        self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.copy_stream):
            dev = self.host.to(self.device, non_blocking=True)
        torch.cuda.current_stream(self.device).wait_stream(self.copy_stream)   # the consumer kernel runs after the transfer
        dev.record_stream(torch.cuda.current_stream(self.device))
        return dev


def load_titan_batch(dataset_path, params: List[Tuple[str, int, str]], dates: List[List[dt.datetime]], stats: Stats,
                     num_input_steps: int, forcing, reader: Optional[NpyPlaneReader] = None, device=None, standardize: bool = True):
    """``Sample.load`` + ``collate_fn`` for the input_output parameters of a batch read from Titan's .npy layout.
    params: (name, level, level_type); dates[b][t]: validity times of sample b.  Returns an ItemBatch on ``device``."""
    from . import datapipe

    B, T = len(dates), len(dates[0])
    paths = [[[titan_plane_path(dataset_path, n, lv, lt, dates[b][t]) for t in range(T)] for b in range(B)] for (n, lv, lt) in params]
    if reader is None:
        with open(paths[0][0][0], "rb") as f:
            shape, _, _ = _read_header(f)
        reader = NpyPlaneReader(shape, len(params), B, T, device=device)
    raw = reader.read(paths)
    names = [f"{n}_{lv}{'m' if lt in ('surface', 'heightAboveGround') else 'hpa'}" for (n, lv, lt) in params]
    return datapipe.load_batch(raw, names, forcing, stats, num_input_steps, standardize)
