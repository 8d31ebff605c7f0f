"""
py4cast plugin module (discovered through its ``py4cast_plugin_`` name prefix, exactly like the
reference's ``py4cast_plugin_example.py``; see py4cast/models.py:23-46): registers the
MI355X-native models.  Put the repository root on PYTHONPATH and ``model_name: HalfUNet`` (or
``HalfUNetMI355X`` next to a real mfai install, which already owns the name ``HalfUNet``)
selects the HIP implementation; likewise ``GraphLam`` / ``GraphLamMI355X`` (mesh GNN on the edge kernels) and
``SwinUNetR`` / ``SwinUNetRMI355X`` (fused window attention).
"""

from dataclasses import dataclass

import torch
from torch import nn

from py4cast_amd.base import ModelABC, ModelType
from py4cast_amd.namedtensor import HAVE_MFAI

from py4cast_amd.halfunet import HalfUNetMI355X, HalfUNetSettings  # noqa: F401,E402

from py4cast_amd.graphlam import GraphLamMI355X, GraphLamSettings  # noqa: F401,E402
from py4cast_amd.swinunetr import SwinUNetRMI355X, SwinUNetRSettings  # noqa: F401,E402
from py4cast_amd.hilam import HiLamMI355X, HiLamSettings  # noqa: F401,E402
from py4cast_amd.hilamparallel import HiLamParallelMI355X, HiLamParallelSettings  # noqa: F401,E402
from py4cast_amd.unetrpp import UNetRPPMI355X, UNetRPPSettings  # noqa: F401,E402

if not HAVE_MFAI:
    # stand-alone: take the upstream names so that config/CLI/model/halfunet.yaml / graphlam.yaml work unchanged
    class HalfUNet(HalfUNetMI355X):
        register = True

    class GraphLAM(GraphLamMI355X):     # mfai's class name = the reference's registry key (tests/test_models.py:145-165)
        register = True

    class GraphLam(GraphLamMI355X):     # the spelling config/CLI/model/graphlam.yaml:2 uses
        register = True

    class SwinUNetR(SwinUNetRMI355X):
        register = True

    class HiLAM(HiLamMI355X):
        register = True

    class HiLAMParallel(HiLamParallelMI355X):
        register = True

    class UNetRPP(UNetRPPMI355X):       # config/CLI/model/unetrpp.yaml:2, tests/test_models.py:160
        register = True


@dataclass
class IdentitySettings:
    name: str = "Identity"


class Identity(ModelABC, nn.Module):
    """
    Same contract as the reference's plugin example (py4cast_plugin_example.py:19-57): keeps the
    first ``out_channels`` features and multiplies by one learnable scalar.  Plain torch ops; it
    exercises the registry and the generic (any nn.Module) rollout path.
    """

    settings_kls = IdentitySettings
    onnx_supported = False
    features_last: bool = True
    supported_num_spatial_dims = (2,)
    num_spatial_dims = 2
    model_type = ModelType.CONVOLUTIONAL
    register: bool = not HAVE_MFAI  # with mfai + the reference's example on the path the name is taken

    def __init__(self, in_channels: int, out_channels: int, input_shape: tuple = None,
                 settings: IdentitySettings = IdentitySettings(), *args, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.input_shape = in_channels, out_channels, input_shape
        self.num_output_features = out_channels
        self.scaler = nn.Parameter(torch.rand(1))
        self._settings = settings
        self.check_required_attributes()

    @property
    def settings(self):
        return self._settings

    def forward(self, x):
        return x[..., : self.num_output_features] * self.scaler
