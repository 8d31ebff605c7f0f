"""
Loss plugins of the hot path, API-compatible with ``py4cast/losses.py`` of the reference
(Py4CastLoss ABC :18-75, WeightedLoss :103-169, ScaledLoss :172-210, CombinedLoss :263-307)
and computed by the HIP kernels of ``csrc/losses.hip`` through the C ABI.

Same constructor/yaml schema (``losses: [{class, weight, params: {loss, reduction}}]``),
same ``prepare(lm, interior_mask, dataset_info)`` / ``forward(prediction, target, mask)``
contract, same output shapes ((B,T), (B,T,*S) with ``reduce_spatial_dim=False``, (B,T,F)),
same error behaviour (unknown torch loss -> NameError; a ScaledLoss member inside a
CombinedLoss raises on the shape mismatch, as the reference's in-place add does).

Differences, on purpose:
* graph-layout tensors (B,T,N,F): the reference raises IndexError (losses.py:156,197 use
  ``dim=(0, 1, 4)``); here the value is defined as the grid-layout result of the same data.
* per-feature weights are uploaded once in ``prepare`` (the reference rebuilds them through an
  lru_cache keyed on (feature names, device), losses.py:77-84).
* ``mask`` may be the marker object returned by ``FusedARLightning.get_mask_on_nan`` (no
  mask tensor is materialised; the kernels derive it from the target's NaNs).
"""

from abc import ABC, abstractmethod
from typing import Optional, Tuple

import torch

from . import _lib as L
from . import ops
from .namedtensor import NamedTensor

SUPPORTED_TORCH_LOSSES = ("MSELoss", "L1Loss")


class _LazyMask:
    """A mask the loss kernels never need as a tensor (lightning.py:787-797 allocates it per step).  The kernels of this
    package recognise the marker and derive the mask in place; for EVERY other consumer -- the reference's plot / metric
    observers (plots.py:524,606; metrics.py:411), arithmetic, indexing, any ``torch.*`` function, attribute access -- the
    object behaves like the reference's literal tensor: it materialises it once, on first use, and forwards to it."""

    _tensor = None

    def _build(self) -> torch.Tensor:  # pragma: no cover - abstract
        raise NotImplementedError

    def materialize(self) -> torch.Tensor:
        if self._tensor is None:
            self._tensor = self._build()
        return self._tensor

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map

        conv = lambda a: a.materialize() if isinstance(a, _LazyMask) else a  # noqa: E731
        return func(*tree_map(conv, args), **tree_map(conv, kwargs or {}))

    def __getattr__(self, name):   # reached only for names the marker itself lacks: shape, dtype, device, sum, to, ...
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)


def _forward_dunder(name):
    def method(self, *args, **kwargs):
        args = tuple(a.materialize() if isinstance(a, _LazyMask) else a for a in args)
        return getattr(self.materialize(), name)(*args, **kwargs)

    method.__name__ = name
    return method


for _n in ("add", "radd", "sub", "rsub", "mul", "rmul", "truediv", "rtruediv", "and", "rand", "or", "ror", "xor", "rxor", "invert",
           "neg", "eq", "ne", "lt", "le", "gt", "ge", "getitem", "len", "iter", "bool", "float", "int", "matmul", "pow", "array"):
    setattr(_LazyMask, f"__{_n}__", _forward_dunder(f"__{_n}__"))
_LazyMask.__hash__ = object.__hash__   # defining __eq__ would otherwise make the markers unhashable


class NanMask(_LazyMask):
    """mask = ~isnan(raw_target) and target = nan_to_num(raw_target), fused in-kernel (lightning.py:792-796).
    ``raw_target`` is the un-masked target tensor."""

    def __init__(self, raw_target: torch.Tensor):
        self.raw_target = raw_target

    def _build(self) -> torch.Tensor:
        return ~torch.isnan(self.raw_target)


class OnesMask(_LazyMask):
    """``torch.ones_like(target)`` (lightning.py:797) without allocating it unless somebody looks."""

    def __init__(self, like: Optional[torch.Tensor] = None):
        self.like = like

    def _build(self) -> torch.Tensor:
        if self.like is None:
            raise RuntimeError("OnesMask() was built without its target: nothing to materialise")
        return torch.ones_like(self.like)


def _mask_spec(mask, target: NamedTensor) -> Tuple[ops.MaskSpec, torch.Tensor]:
    """(mask specification for the kernels, target tensor they read).  ``target.tensor`` is only touched when it is needed: with a
    NanMask marker the kernels read the RAW target and a lazily masked target (lightning._LazyMaskedTarget) stays unbuilt."""
    if isinstance(mask, NanMask):
        return ops.MaskSpec(L.MASK_FROM_NAN), mask.raw_target
    if mask is None or isinstance(mask, OnesMask):
        return ops.MaskSpec(L.MASK_NONE), target.tensor
    return ops.MaskSpec.from_tensor(mask), target.tensor


class Py4CastLoss(ABC):
    """losses.py:18-75."""

    def __init__(self, loss: str, *args, **kwargs) -> None:
        if hasattr(torch.nn, loss):
            self.loss = getattr(torch.nn, loss)(*args, **kwargs)  # kept for introspection / repr parity
        elif loss in globals():
            self.loss = globals()[loss](*args, **kwargs)
        else:
            raise NameError(f"Loss: {loss} is not defined")
        self.loss_name = loss
        reduction = kwargs.get("reduction", "mean")
        if hasattr(torch.nn, loss) and reduction != "none":
            # the reference would broadcast a scalar through the weighted sums; py4cast's yaml
            # always sets "none" (halfunet.yaml:3-8).  Refuse rather than silently differ.
            raise ValueError(f'{type(self).__name__}: reduction must be "none" (got {reduction!r})')

    @property
    def kind(self) -> int:
        """Kernel code of the loss (MSELoss / L1Loss: the fused HIP path); other torch losses take the generic torch-op path and the
        fused rollout asks `fused_capable` first."""
        return ops.loss_kind_code(self.loss_name)

    @property
    def fused_capable(self) -> bool:
        return self.loss_name in SUPPORTED_TORCH_LOSSES

    @abstractmethod
    def prepare(self, lm, interior_mask: torch.Tensor, dataset_info) -> None:
        """Prepare the loss function using the dataset informations and the interior mask."""

    @abstractmethod
    def forward(self, prediction: NamedTensor, target: NamedTensor, mask) -> torch.Tensor:
        """Compute the loss function."""

    def register_loss_state_buffers(self, lm, interior_mask: torch.Tensor, loss_state_weight: dict,
                                    squeeze_mask: bool = False) -> None:
        """losses.py:52-72: registers interior_mask(_s) on the module and counts interior points."""
        self.loss_state_weight = loss_state_weight
        attr_name = "interior_mask_s" if squeeze_mask else "interior_mask"
        if not hasattr(lm, attr_name):
            lm.register_buffer(attr_name, interior_mask.squeeze(-1) if squeeze_mask else interior_mask, persistent=False)
        self.num_interior = torch.sum(interior_mask).item()
        self._weights_cache = {}

    def __call__(self, *args, **kwds):
        return self.forward(*args, **kwds)

    def weights(self, feature_names: Tuple[str], device: torch.device) -> torch.Tensor:
        """losses.py:77-84 (per (feature names, device) cache, built once)."""
        key = (tuple(feature_names), str(device))
        w = self._weights_cache.get(key)
        if w is None:
            w = torch.stack(
                [torch.as_tensor(self.loss_state_weight[name], dtype=torch.float32) for name in feature_names]
            ).to(device)
            self._weights_cache[key] = w
        return w

    def _interior_flat(self, lm, device) -> torch.Tensor:
        im = getattr(lm, "interior_mask")
        flat = getattr(self, "_interior_flat_cache", None)
        if flat is None or flat.device != device or flat.numel() != im.numel():
            flat = im.detach().reshape(-1).to(device=device, dtype=torch.float32).contiguous()
            self._interior_flat_cache = flat
        return flat


class WeightedLoss(Py4CastLoss):
    """losses.py:103-169: per-feature weight state_weight/diff_std**p, masked spatial mean -> (B,T)."""

    def prepare(self, lm, interior_mask: torch.Tensor, dataset_info) -> None:
        exponent = 2.0 if self.loss_name == "MSELoss" else 1.0  # losses.py:119
        loss_state_weight = {}
        for name in dataset_info.state_weights:
            loss_state_weight[name] = dataset_info.state_weights[name] / (
                dataset_info.diff_stats[name]["std"] ** exponent
            )
        self.register_loss_state_buffers(lm, interior_mask, loss_state_weight, squeeze_mask=True)
        self.lm = lm

    def forward(self, prediction: NamedTensor, target: NamedTensor, mask, reduce_spatial_dim: bool = True,
                masked_count: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.loss_name not in SUPPORTED_TORCH_LOSSES:
            return self._forward_generic(prediction, target, mask, reduce_spatial_dim)
        spec, tgt = _mask_spec(mask, target)
        weights = self.weights(tuple(prediction.feature_names), prediction.device)
        if not reduce_spatial_dim:
            return ops.weighted_loss_map(prediction.tensor, tgt, spec, weights, self.kind)
        interior = self._interior_flat(self.lm, prediction.device)
        return ops.weighted_loss(prediction.tensor, tgt, spec, weights, interior, self.num_interior, self.kind,
                                 count=masked_count)


    def _forward_generic(self, prediction, target, mask, reduce_spatial_dim):
        """Any other element-wise ``torch.nn`` loss with ``reduction="none"`` (SmoothL1Loss, HuberLoss, ...; losses.py:25-31 accepts
        every name torch.nn has): the reference's op sequence (losses.py:143-169) on the device with torch ops -- unfused, differentiable
        through autograd; the HIP kernels implement the two losses the shipped configurations use."""
        m, tgt = _dense_mask(mask, target)
        torch_loss = self.loss(prediction.tensor * m, tgt * m)
        weights = self.weights(tuple(prediction.feature_names), prediction.device)
        weighted = torch.sum(torch_loss * weights, dim=-1)
        if not reduce_spatial_dim:
            return weighted
        union = torch.any(m.bool() if m.dtype != torch.bool else m, dim=(0, 1, m.dim() - 1))
        interior = getattr(self.lm, "interior_mask_s").to(weighted.device)
        spatial = tuple(range(2, weighted.dim()))
        return torch.sum(weighted * interior.reshape(weighted.shape[2:]), dim=spatial) / (self.num_interior - (~union).sum())


def _dense_mask(mask, target: NamedTensor):
    """(mask tensor broadcastable against the target, target tensor) for the torch-op paths: the markers are materialised."""
    if isinstance(mask, NanMask):
        raw = mask.raw_target
        return ~torch.isnan(raw), torch.nan_to_num(raw)
    if mask is None or isinstance(mask, OnesMask):
        return torch.ones_like(target.tensor, dtype=torch.bool), target.tensor
    return mask, target.tensor


class ScaledLoss(Py4CastLoss):
    """losses.py:172-210: per-feature masked spatial mean, sqrt if MSE, times std -> (B,T,F)."""

    def prepare(self, lm, interior_mask: torch.Tensor, dataset_info) -> None:
        loss_state_weight = {}
        for name in dataset_info.state_weights:
            loss_state_weight[name] = dataset_info.stats[name]["std"]
        self.register_loss_state_buffers(lm, interior_mask, loss_state_weight)
        self.lm = lm

    def forward(self, prediction: NamedTensor, target: NamedTensor, mask) -> torch.Tensor:
        if self.loss_name not in SUPPORTED_TORCH_LOSSES:   # generic torch-op path, losses.py:195-210 (no sqrt: that is MSELoss only)
            m, tgt = _dense_mask(mask, target)
            torch_loss = self.loss(prediction.tensor * m, tgt * m)
            union = torch.any(m.bool() if m.dtype != torch.bool else m, dim=(0, 1, m.dim() - 1))
            interior = getattr(self.lm, "interior_mask").to(torch_loss.device)
            spatial = tuple(range(2, torch_loss.dim() - 1))
            mean_loss = torch.sum(torch_loss * interior.reshape(torch_loss.shape[2:-1] + (1,)), dim=spatial) / (self.num_interior - (~union).sum())
            return mean_loss * self.weights(tuple(prediction.feature_names), prediction.device)
        spec, tgt = _mask_spec(mask, target)
        std = self.weights(tuple(prediction.feature_names), prediction.device)
        interior = self._interior_flat(self.lm, prediction.device)
        return ops.scaled_loss(prediction.tensor, tgt, spec, std, interior, self.num_interior, self.kind)


class PerceptualLossPy4Cast(Py4CastLoss):
    """losses.py:213-260 wraps mfai's PerceptualLoss (a VGG-style network): out of the hot-path scope."""

    def __init__(self, *args, **kwargs) -> None:
        raise NotImplementedError(
            "PerceptualLossPy4Cast needs mfai.pytorch.losses.perceptual.PerceptualLoss; it is outside the "
            "MI355X hot-path scope (SURVEY.md section 2, row 2)."
        )

    def prepare(self, lm, interior_mask, dataset_info) -> None:  # pragma: no cover
        pass

    def forward(self, prediction, target, mask):  # pragma: no cover
        pass


class CombinedLoss(Py4CastLoss):
    """losses.py:263-307: weighted sum of Py4CastLoss members built from yaml dicts."""

    def __init__(self, losses_config: list):
        self.losses = []
        for loss_conf in losses_config:
            LossClass = globals()[loss_conf["class"]]  # string class names, as in the reference (:271)
            weight = loss_conf.get("weight", 1.0)
            kwargs = loss_conf.get("params", {})
            self.losses.append((LossClass(**kwargs), weight))

    def prepare(self, lm, interior_mask: torch.Tensor, dataset_info):
        for loss, _ in self.losses:
            if hasattr(loss, "prepare"):
                loss.prepare(lm, interior_mask, dataset_info)

    def forward(self, prediction: NamedTensor, target: NamedTensor, mask, **kwargs) -> torch.Tensor:
        loss_shape = (
            prediction.tensor.shape[:2] if kwargs.get("reduce_spatial_dim", True) else prediction.tensor.shape[:-1]
        )
        total_loss = torch.zeros(loss_shape, device=prediction.tensor.device)
        if len(self.losses) > 1 and "masked_count" not in kwargs and kwargs.get("reduce_spatial_dim", True):
            # one union-mask pass shared by every member instead of one per member
            spec, tgt = _mask_spec(mask, target)
            if all(isinstance(l, WeightedLoss) for l, _ in self.losses):
                kwargs = dict(kwargs, masked_count=ops.masked_count(spec, tgt))
        for loss, weight in self.losses:
            total_loss += weight * loss(prediction, target, mask, **kwargs)
        return total_loss

    def __repr__(self):
        return "CombinedLoss(" + ", ".join(f"{w}*{type(l).__name__}({l.loss_name})" for l, w in self.losses) + ")"
