"""
py4cast_amd -- MI355X (gfx950) implementation of the py4cast autoregressive training hot
path: AR rollout (py4cast/lightning.py:495-797), losses (py4cast/losses.py) and the model
kernels behind py4cast's plugin registry (py4cast/models.py), as hand-written HIP kernels
in ``libpy4cast_hip.so`` (C ABI: include/py4cast_hip.h).  Host code is Python on
PyTorch-ROCm; PyTorch provides device memory, streams and torch.distributed only.
"""

__version__ = "0.1.0"

import os as _os

# Library convolutions (only the UNETR decoder of SwinUNetR uses them): MIOpen's default find mode benchmarks every
# applicable solver on first use of a shape, and on this stack (ROCm 7.2 / MIOpen 3.5, gfx950) one of the candidates it tries for
# the decoder's backward convolutions takes the process down with a GPU memory access fault (reproduced with plain torch modules
# and an empty MIOpen cache; scratch notes in DESIGN.md section 8).  FAST mode (immediate-mode heuristics, no benchmarking pass)
# avoids it.  Respect an explicit user choice.
_os.environ.setdefault("MIOPEN_FIND_MODE", "2")

# Library GEMMs (Linear layers, token-axis projections, pixel-block convolutions of SwinUNetR / UNetRPP): hipBLASLt's default
# heuristic picks deep-K stream-K kernels for the skinny shapes of the early Swin stages (K = 24..96 against 10^5 rows) that run at a
# few hundred GB/s.  PyTorch's TunableOp selects per shape among the library's own solutions; the selections for the bench
# workloads were tuned once on an MI355X (tools/diagnostics/tune_gemms.sh -> merge_tunable.py) and ship as
# tuning/tunableop_gfx950.csv, loaded with tuning switched OFF (a lookup per GEMM shape; unknown shapes take the default; the file's
# validator lines make the library ignore it on another ROCm / hipBLASLt / GPU).  SwinUNetR 55.1 -> 49 ms per step.
# Here: environment only (torch reads it at its first GEMM; nothing at import touches the device); the file itself is handed to
# torch.cuda.tunable.read_file by the first native call on a GPU tensor (_lib.require_cuda) -- TunableOp's own file-name variable
# inserts the device ordinal into the name, which would need one copy per rank.  Respect an explicit user choice (PYTORCH_TUNABLEOP_ENABLED set either way), and P4C_NO_TUNED_GEMMS=1.
_TUNED_GEMMS = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tuning", "tunableop_gfx950.csv")
# (round 3: nothing is switched on at import any more -- importing this package used to set PYTORCH_TUNABLEOP_ENABLED for the
# whole host process; now _lib._load_tuned_gemms enables the lookup itself, tuning and file writing off, at that first call)
if ("PYTORCH_TUNABLEOP_ENABLED" not in _os.environ and _os.environ.get("P4C_NO_TUNED_GEMMS") != "1"
        and _os.path.exists(_TUNED_GEMMS)):
    _os.environ["P4C_TUNED_GEMMS_FILE"] = _TUNED_GEMMS   # read by _lib.require_cuda at the first native call on a GPU tensor


def invalidate_param_caches():
    """Parameter storage was written behind autograd's back (``p.data.copy_``, a collective into ``p.data``): drop what the ops
    cached per parameter version.  See INTEGRATION.md, "Parameter caches"."""
    from . import _lib

    _lib.invalidate_param_caches()
