"""
Model plugin registry -- same contract as ``py4cast/models.py`` of the reference (:16-89).

* starts from ``mfai.pytorch.models.registry`` when mfai is installed (minus the two models
  whose forward takes more than ``x``, models.py:19-20);
* auto-discovers every importable top-level module named ``py4cast_plugin_*`` and registers
  the ``ModelABC`` subclasses with ``register = True`` (models.py:23-46); a duplicate class
  name raises ``ValueError`` (models.py:42-45);
* ``get_model_kls_and_settings`` / ``build_model_from_settings`` keep the reference's
  signatures and the positional constructor call (models.py:50-89).

The MI355X models live in the plugin module ``py4cast_plugin_mi355x`` (repo root), exactly
where a third-party plugin would: nothing here imports them directly.
"""

import importlib
import pkgutil
from typing import Any, Tuple

from .base import ModelABC

registry = {}
try:  # pragma: no cover - mfai absent in the build image
    from mfai.pytorch.models import registry as mfai_registry  # type: ignore

    registry.update(mfai_registry)
    registry.pop("PanguWeather", None)
    registry.pop("ArchesWeather", None)
except Exception:
    pass

PLUGIN_PREFIX = "py4cast_plugin_"


def _discover():
    discovered = {}
    for _, name, _ in pkgutil.iter_modules():
        if name.startswith(PLUGIN_PREFIX):
            try:
                discovered[name] = importlib.import_module(name)
            except ImportError as e:  # a plugin whose own dependencies are missing must not break the others
                import warnings

                warnings.warn(f"py4cast plugin {name} could not be imported: {e}")
    return discovered


discovered_modules = _discover()

for module_name, module in discovered_modules.items():
    for name, kls in list(module.__dict__.items()):
        if isinstance(kls, type) and issubclass(kls, ModelABC) and kls != ModelABC and kls.register:
            if kls.__name__ in registry:
                raise ValueError(f"Model {kls.__name__} from plugin {module_name} already exists in the registry.")
            registry[kls.__name__] = kls
all_nn_architectures = list(registry)


def get_model_kls_and_settings(model_name: str, settings: dict):
    """Returns the classes for a model and its settings instance (models.py:50-63)."""
    try:
        model_kls = registry[model_name]
    except KeyError as e:
        raise KeyError(
            f"Model {model_name} not found in registry of {__file__}. Did you add it ? Names are {registry.keys()}"
        ) from e
    settings_kls = model_kls.settings_kls
    model_settings = settings_kls(**settings)
    return model_kls, model_settings


def build_model_from_settings(
    network_name: str, num_input_features: int, num_output_features: int, settings: dict, input_shape: tuple,
    *args, **kwargs,
) -> Tuple[ModelABC, Any]:
    """Instanciates a model based on its name and an optional settings dict (models.py:66-89)."""
    model_kls, model_settings = get_model_kls_and_settings(network_name, settings)
    return (
        model_kls(num_input_features, num_output_features, input_shape, model_settings, *args, **kwargs),
        model_settings,
    )
