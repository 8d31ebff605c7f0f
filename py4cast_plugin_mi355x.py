"""
py4cast plugin module (discovered through its ``py4cast_plugin_`` name prefix, exactly like the
reference's ``py4cast_plugin_example.py``; see py4cast/models.py:23-46): registers the
MI355X-native models.  Put the repository root on PYTHONPATH and ``model_name: HalfUNet`` (or
``HalfUNetMI355X`` next to a real mfai install, which already owns the name ``HalfUNet``)
selects the HIP implementation.
"""

from py4cast_amd.namedtensor import HAVE_MFAI

try:
    from py4cast_amd.halfunet import HalfUNetMI355X  # noqa: F401

    if not HAVE_MFAI:
        # stand-alone: take the upstream name so that config/CLI/model/halfunet.yaml works unchanged
        class HalfUNet(HalfUNetMI355X):
            register = True
except ImportError:  # pragma: no cover
    raise
