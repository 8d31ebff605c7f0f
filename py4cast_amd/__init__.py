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
