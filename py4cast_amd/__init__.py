"""
py4cast_amd -- MI355X (gfx950) implementation of the py4cast autoregressive training hot
path: AR rollout (py4cast/lightning.py:495-797), losses (py4cast/losses.py) and the model
kernels behind py4cast's plugin registry (py4cast/models.py), as hand-written HIP kernels
in ``libpy4cast_hip.so`` (C ABI: include/py4cast_hip.h).  Host code is Python on
PyTorch-ROCm; PyTorch provides device memory, streams and torch.distributed only.
"""

__version__ = "0.1.0"
