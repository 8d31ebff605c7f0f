"""
HalfUNet on MI355X: host-side wrapper (a py4cast model plugin) around the HIP plan
``p4c_halfunet_forward / p4c_halfunet_backward`` (csrc/halfunet.cpp).

Plugin contract: py4cast/models.py:23-89 and doc/add_features_contribute.md:19-30 of the
reference -- ``ModelABC`` + ``nn.Module``, positional ctor ``(in_channels, out_channels,
input_shape, settings)``, class attributes below.  The architecture and the parameter / buffer
names follow mfai's ``HalfUNet`` (``encoder1.enc1conv1.weight`` ...; restated in
oracle/halfunet.py, parity unpinned because mfai is absent from the reference checkout), so
state_dicts are interchangeable with a torch-native HalfUNet.

``features_last = True``: the model consumes and produces (B,H,W,C) tensors, the layout of the
reference's NamedTensors, so the rollout hands its tensors over without a permute
(lightning.py:591-596 permutes only for ``features_second`` models).  mfai's own HalfUNet
is NCHW; the registry contract allows either.
"""

import ctypes
import os
import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import nn

from . import _lib as L
from ._lib_model import HalfUNetDesc
from .base import ModelABC, ModelType

NF = 64
BLOCKS = ("enc1", "enc2", "enc3", "enc4", "enc5", "decoder")
BLOCK_ATTR = ("encoder1", "encoder2", "encoder3", "encoder4", "encoder5", "decoder")


@dataclass
class HalfUNetSettings:
    """mfai's HalfUNetSettings fields (config/CLI/model/halfunet.yaml:19-26) + the MI355X knobs."""

    num_filters: int = 64
    dilation: int = 1
    bias: bool = False
    use_ghost: bool = False
    last_activation: str = "Identity"
    absolute_pos_embed: bool = False
    autopad_enabled: bool = False   # mfai's AutoPaddingModel: grids that are not a multiple of 16 are zero-padded (centred) and cropped
    # MI355X-specific
    norm: str = "batch"  # "batch" (mfai) or "group" (GroupNorm, BASELINE.json north star)
    groups: int = 8
    compute_dtype: str = "f32"  # matrix-core input type: "f32" (exact fp32 MFMA) or "bf16" (autocast-like)
    activation_dtype: Optional[str] = None  # HBM storage of activations / their gradients: "f32" | "bf16"; None = compute_dtype


def pad32(c: int) -> int:
    return (c + 31) // 32 * 32


class _Holder(nn.Module):
    """Parameter/buffer container giving mfai's dotted names."""


class _HalfUNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, training, *params):
        L.require_cuda(x)
        B, H, W, C = x.shape
        assert C == model.cin_pad and x.dtype == model.act_dtype and x.is_contiguous()
        desc = model._desc(B, H, W)
        flat = model._flat_params()
        saved_bytes, scratch = model._workspaces(desc, x.device)
        saved = torch.empty(saved_bytes, dtype=torch.uint8, device=x.device)
        y = torch.empty(B, H, W, NF, dtype=model.act_dtype, device=x.device)
        L.call("p4c_halfunet_forward", ctypes.byref(desc), L.ptr(x), L.ptr(flat), L.ptr(model._running), L.ptr(y),
               L.ptr(saved), L.ptr(scratch), int(training), L.stream(x.device))
        ctx.model, ctx.desc, ctx.training = model, desc, training
        ctx.save_for_backward(x, saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, saved = ctx.saved_tensors
        model, desc = ctx.model, ctx.desc
        dy = dy.contiguous().to(model.act_dtype)
        flat = model._flat_params()
        _, scratch = model._workspaces(desc, x.device)
        gflat = torch.zeros_like(flat)
        dx = torch.empty_like(dy) if desc.dx_channels > 0 else None
        L.call("p4c_halfunet_backward", ctypes.byref(desc), L.ptr(x), L.ptr(flat), L.ptr(dy), L.ptr(dx), L.ptr(gflat),
               L.ptr(saved), L.ptr(scratch), int(ctx.training), L.stream(x.device))
        dxp = None
        if dx is not None:
            dxp = torch.zeros_like(x)
            dxp[..., : desc.dx_channels] = dx[..., : desc.dx_channels]
        grads = tuple(gflat[o : o + n].view(s) for (o, n, s) in model._param_slices)
        return (dxp, None, None) + grads


class HalfUNetMI355X(ModelABC, nn.Module):
    settings_kls = HalfUNetSettings
    onnx_supported = False
    supported_num_spatial_dims = (2,)
    num_spatial_dims = 2
    features_last = True
    model_type = ModelType.CONVOLUTIONAL
    register = True
    is_native_hip = True  # common_step: precision is handled by the kernels, no torch.autocast

    def __init__(self, in_channels: int, out_channels: int, input_shape: Optional[Tuple[int, int]] = None,
                 settings: HalfUNetSettings = HalfUNetSettings(), *args, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.input_shape = in_channels, out_channels, input_shape
        self._settings = settings
        unsupported = []
        if settings.num_filters % 2 or settings.num_filters < 2:
            unsupported.append(f"num_filters={settings.num_filters} (must be even)")
        if settings.dilation < 1:
            unsupported.append(f"dilation={settings.dilation}")
        if not hasattr(nn, settings.last_activation):
            unsupported.append(f"last_activation={settings.last_activation}")
        if settings.absolute_pos_embed:
            unsupported.append("absolute_pos_embed=True")
        if settings.norm not in ("batch", "group"):
            unsupported.append(f"norm={settings.norm}")
        if settings.compute_dtype not in ("f32", "bf16"):
            unsupported.append(f"compute_dtype={settings.compute_dtype}")
        act = settings.activation_dtype or settings.compute_dtype
        if act not in ("f32", "bf16") or (act == "bf16" and settings.compute_dtype != "bf16"):
            unsupported.append(f"activation_dtype={settings.activation_dtype} with compute_dtype={settings.compute_dtype}")
        # the fused plan (p4c_halfunet_forward / _backward) is built for the yaml's network: 64 filters, no dilation, no bias, no
        # last activation; every other setting of mfai's HalfUNetSettings runs the module path below (`_forward_modules`)
        self.module_path = bool(settings.use_ghost or settings.num_filters != NF or settings.dilation != 1 or settings.bias
                                or settings.last_activation != "Identity")
        if not self.module_path:
            if out_channels > NF:
                unsupported.append(f"out_channels={out_channels} > 64")
            if pad32(in_channels) > 96:
                unsupported.append(f"in_channels={in_channels} > 96")
        if unsupported:
            raise NotImplementedError("HalfUNetMI355X: unsupported settings: " + ", ".join(unsupported))
        self.cin_pad = pad32(in_channels)
        self.dx_channels = min(in_channels, NF)  # gradient wrt the leading (previous-state) channels
        self.compute_dtype = torch.float32 if settings.compute_dtype == "f32" else torch.bfloat16
        self.act_dtype = torch.float32 if act == "f32" else torch.bfloat16  # dtype of x / y / dy / dx handed to the plan
        self.timed_entry_points = ("p4c_halfunet_forward", "p4c_halfunet_backward", "p4c_build_x",
                                   "p4c_ar_update_loss_fwd", "p4c_ar_update_loss_fwd_next", "p4c_ar_update_loss_bwd")

        self.use_ghost = bool(settings.use_ghost)
        if self.module_path:
            self._init_modules(in_channels, out_channels, settings)
            return
        # parameters in the order of p4c_halfunet_param_count (include/py4cast_hip.h)
        self._param_slices = []
        off = 0
        norm_bufs = []
        for bi, (blk, attr) in enumerate(zip(BLOCKS, BLOCK_ATTR)):
            holder = _Holder()
            for j in (1, 2):
                cin = in_channels if (bi == 0 and j == 1) else NF
                conv, norm = _Holder(), _Holder()
                w = torch.empty(NF, cin, 3, 3)
                nn.init.kaiming_uniform_(w, a=5**0.5)  # nn.Conv2d default init
                conv.weight = nn.Parameter(w)
                norm.weight = nn.Parameter(torch.ones(NF))
                norm.bias = nn.Parameter(torch.zeros(NF))
                if settings.norm == "batch":
                    norm.register_buffer("running_mean", torch.zeros(NF))
                    norm.register_buffer("running_var", torch.ones(NF))
                    norm.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
                norm_bufs.append(norm)
                setattr(holder, f"{blk}conv{j}", conv)
                setattr(holder, f"{blk}norm{j}", norm)
                for p in (conv.weight, norm.weight, norm.bias):
                    self._param_slices.append((off, p.numel(), tuple(p.shape)))
                    off += p.numel()
            setattr(self, attr, holder)
        self.outconv = _Holder()
        w = torch.empty(out_channels, NF, 1, 1)
        nn.init.kaiming_uniform_(w, a=5**0.5)
        self.outconv.weight = nn.Parameter(w)
        self._param_slices.append((off, w.numel(), tuple(w.shape)))
        off += w.numel()
        self._nparams = off
        self._norms = norm_bufs
        self._flat = None
        self._running = None
        self._scratch = {}
        self.check_required_attributes()

    # ---------------------------------------------------------------- module path: Ghost blocks and the non-default yaml settings
    def _init_modules(self, in_channels, out_channels, settings):
        """Every HalfUNetSettings combination the fused plan is not built for (halfunet.yaml:19-26: ``use_ghost``, ``num_filters``
        other than 64, ``dilation``, ``bias``, ``last_activation``), as a network of per-layer nodes with mfai's parameter names
        (restated in oracle/halfunet.py): ``encoder1.enc1conv1.{weight,bias}`` / ``enc1norm1.*``, or for Ghost blocks
        ``encoder1.enc1ghost1.conv.*`` / ``.sepconv.*`` / ``.bn.*``; ``outconv.{weight,bias}``.
        Convolutions without dilation to <= 64 channels from <= 96 run on the MFMA conv kernels (ops_model.conv_nhwc: forward, data
        and weight gradient; fewer than 64 output channels as zero rows of a 64-channel launch), the Ghost "cheap operation" at 64
        filters on csrc/depthwise.hip (ops_ghost.ghost_dw); wider or dilated convolutions go through the library; normalisation,
        pooling and up-sampling are torch ops on the features-last tensors.  The fused C++ plan (p4c_halfunet_forward) serves the
        yaml's network only, so the rollout takes the generic per-step path here."""
        nf = settings.num_filters

        def conv_holder(co, ci, ks, groups=1):
            h = _Holder()
            w = torch.empty(co, ci // groups, ks, ks)
            nn.init.kaiming_uniform_(w, a=5**0.5)
            h.weight = nn.Parameter(w)
            if settings.bias:   # nn.Conv2d's default bias initialisation
                bound = 1.0 / math.sqrt(ci // groups * ks * ks)
                h.bias = nn.Parameter(torch.empty(co).uniform_(-bound, bound))
            return h

        def norm_holder():
            n = _Holder()
            n.weight, n.bias = nn.Parameter(torch.ones(nf)), nn.Parameter(torch.zeros(nf))
            if settings.norm == "batch":
                n.register_buffer("running_mean", torch.zeros(nf))
                n.register_buffer("running_var", torch.ones(nf))
                n.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
            return n

        for blk, attr in zip(BLOCKS, BLOCK_ATTR):
            holder = _Holder()
            for j in (1, 2):
                cin = in_channels if (attr == "encoder1" and j == 1) else nf
                if self.use_ghost:
                    g = _Holder()
                    g.conv, g.sepconv, g.bn = conv_holder(nf // 2, cin, 3), conv_holder(nf // 2, nf // 2, 3, groups=nf // 2), norm_holder()
                    setattr(holder, f"{blk}ghost{j}", g)
                else:
                    setattr(holder, f"{blk}conv{j}", conv_holder(nf, cin, 3))
                    setattr(holder, f"{blk}norm{j}", norm_holder())
            setattr(self, attr, holder)
        self.outconv = conv_holder(out_channels, nf, 1)
        self.activation = getattr(nn, settings.last_activation)()
        self.native_rollout = None          # instance attribute shadows the method: AutoRegressiveLightning takes the generic path
        self.check_required_attributes()

    def _conv_module(self, c, x, dilation=1):
        """"same" convolution of a features-last tensor with the holder's weight (+ bias)."""
        from . import ops_model as OM

        F = torch.nn.functional
        w = c.weight
        if dilation == 1 and OM.conv_nhwc_supported(x, w):
            y = OM.conv_nhwc(x, w)
        else:
            pad = dilation * (w.shape[-1] // 2)
            y = F.conv2d(x.permute(0, 3, 1, 2), w.to(x.dtype), None, padding=pad, dilation=dilation).permute(0, 2, 3, 1)
        b = getattr(c, "bias", None)
        return y if b is None else y + b.to(y.dtype)

    def _norm_relu(self, n, y):
        F = torch.nn.functional
        v = y.permute(0, 3, 1, 2).float()                                   # NCHW-shaped view of features-last memory
        if self._settings.norm == "batch":
            v = F.batch_norm(v, n.running_mean, n.running_var, n.weight, n.bias, self.training, 0.1, 1e-5)
            if self.training:
                n.num_batches_tracked += 1
        else:
            v = F.group_norm(v, self._settings.groups, n.weight, n.bias, 1e-5)
        return F.relu(v).to(y.dtype).permute(0, 2, 3, 1)

    def _ghost_module(self, g, x):
        from . import ops_ghost as OG
        from . import ops_model as OM

        F = torch.nn.functional
        nf, dil = self._settings.num_filters, self._settings.dilation
        if nf == NF and dil == 1 and OM.conv_nhwc_supported(x, g.conv.weight):
            w = F.pad(g.conv.weight, (0, 0, 0, 0, 0, 0, 0, NF // 2))        # 32 real output channels of a 64-channel launch
            y = OM.conv_nhwc(x, w)
            if getattr(g.conv, "bias", None) is not None:
                y = y + F.pad(g.conv.bias, (0, NF // 2)).to(y.dtype)
            y = OG.ghost_dw(y, g.sepconv.weight)                            # (B,H,W,64): [primary | depthwise]
            if getattr(g.sepconv, "bias", None) is not None:
                y = y + F.pad(g.sepconv.bias, (NF // 2, 0)).to(y.dtype)
        else:
            prim = self._conv_module(g.conv, x, dil)                        # (B,H,W,nf/2)
            cheap = F.conv2d(prim.permute(0, 3, 1, 2), g.sepconv.weight.to(prim.dtype),
                             None if getattr(g.sepconv, "bias", None) is None else g.sepconv.bias.to(prim.dtype), padding=1,
                             groups=nf // 2).permute(0, 2, 3, 1)
            y = torch.cat([prim, cheap], dim=-1)
        return self._norm_relu(g.bn, y)

    def _forward_modules(self, x):
        F = torch.nn.functional
        out_dtype = x.dtype
        x = x.contiguous().to(self.act_dtype)
        dil = self._settings.dilation

        def block(blk, holder, h):
            if self.use_ghost:
                return self._ghost_module(getattr(holder, f"{blk}ghost2"), self._ghost_module(getattr(holder, f"{blk}ghost1"), h))
            for j in (1, 2):
                h = self._norm_relu(getattr(holder, f"{blk}norm{j}"), self._conv_module(getattr(holder, f"{blk}conv{j}"), h, dil))
            return h

        levels = []
        h = x
        for k, (blk, attr) in enumerate(zip(BLOCKS[:5], BLOCK_ATTR[:5])):
            if k > 0:
                h = F.max_pool2d(h.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
            h = block(blk, getattr(self, attr), h)
            levels.append(h)
        s = levels[0].float()
        for k in range(1, 5):
            up = F.interpolate(levels[k].permute(0, 3, 1, 2).float(), scale_factor=2**k, mode="bilinear", align_corners=False)
            s = s + up.permute(0, 2, 3, 1)
        d = block("decoder", self.decoder, s.to(self.act_dtype))
        y = self._conv_module(self.outconv, d)
        if not isinstance(self.activation, nn.Identity):
            y = self.activation(y.float()).to(y.dtype)
        return y if y.dtype == out_dtype or not out_dtype.is_floating_point else y.to(out_dtype)

    @property
    def settings(self):
        return self._settings

    # ---------------------------------------------------------------- flat views
    def _ordered_params(self):
        out = []
        for blk, attr in zip(BLOCKS, BLOCK_ATTR):
            holder = getattr(self, attr)
            for j in (1, 2):
                conv, norm = getattr(holder, f"{blk}conv{j}"), getattr(holder, f"{blk}norm{j}")
                out += [conv.weight, norm.weight, norm.bias]
        out.append(self.outconv.weight)
        return out

    def _flat_params(self) -> torch.Tensor:
        """One flat fp32 buffer holding every parameter (the named nn.Parameters are views of it)."""
        params = self._ordered_params()
        dev = params[0].device
        ok = self._flat is not None and self._flat.device == dev
        if ok:
            base = self._flat.data_ptr()
            for p, (o, n, _) in zip(params, self._param_slices):
                if p.data_ptr() != base + 4 * o:
                    ok = False
                    break
        if not ok and params[0].is_cuda:
            # somebody else (trainer.FlatDDP in sharded mode) may already have laid the parameters out flat, in this order: adopt
            from .optim import _flat_view

            adopt = _flat_view([p.data for p in params])
            if adopt is not None and adopt.numel() == self._nparams:
                self._flat = adopt
                return adopt
        if not ok:  # first use, or the module was moved / a state_dict replaced storages: re-flatten
            flat = torch.empty(self._nparams, dtype=torch.float32, device=dev)
            for p, (o, n, s) in zip(params, self._param_slices):
                flat[o : o + n].copy_(p.data.reshape(-1))
                p.data = flat[o : o + n].view(s)
            self._flat = flat
        return self._flat

    def _flat_grad_target(self):
        """The flat fp32 buffer every ``param.grad`` is a view of, in parameter order (the layout FlatDDP sets up), or
        None.  When it exists the backward plan accumulates straight into it (the C entry point is ``+=`` anyway) and
        autograd is handed no per-parameter gradients: that saves one zero-fill and 37 tiny ``grad +=`` launches."""
        params = self._ordered_params()
        g0 = params[0].grad
        if g0 is None or g0.dtype != torch.float32 or not g0.is_contiguous():
            return None
        base = g0.data_ptr() - 4 * self._param_slices[0][0]
        for p, (o, n, _) in zip(params, self._param_slices):
            g = p.grad
            if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() != base + 4 * o:
                return None
        store = g0.untyped_storage()
        first = (base - store.data_ptr()) // 4
        if base < store.data_ptr() or (first + self._nparams) * 4 > store.nbytes():
            return None
        return torch.empty(0, dtype=torch.float32, device=g0.device).set_(store, first, (self._nparams,))

    def _running_stats(self, device) -> torch.Tensor:
        """[12][2][64] flat running statistics, aliased by the named BatchNorm buffers."""
        if self._settings.norm != "batch":
            if self._running is None or self._running.device != device:
                self._running = torch.zeros(12 * 128, device=device)
            return self._running
        ok = self._running is not None and self._running.device == device
        if ok:
            base = self._running.data_ptr()
            for i, nb in enumerate(self._norms):
                if nb.running_mean.data_ptr() != base + 4 * (i * 128) or nb.running_var.data_ptr() != base + 4 * (i * 128 + 64):
                    ok = False
                    break
        if not ok:
            r = torch.empty(12 * 128, dtype=torch.float32, device=device)
            for i, nb in enumerate(self._norms):
                r[i * 128 : i * 128 + 64].copy_(nb.running_mean)
                r[i * 128 + 64 : i * 128 + 128].copy_(nb.running_var)
                nb.running_mean = r[i * 128 : i * 128 + 64]
                nb.running_var = r[i * 128 + 64 : i * 128 + 128]
            self._running = r
        return self._running

    def _desc(self, B, H, W) -> HalfUNetDesc:
        if H % 16 or W % 16:
            raise L.P4CError(f"HalfUNetMI355X: grid {H}x{W} must be a multiple of 16 in both dimensions")
        s = self._settings
        return HalfUNetDesc(B, H, W, self.in_channels, self.cin_pad, self.out_channels, self.dx_channels,
                            L.dtype_code(self.act_dtype),
                            0 if s.norm == "batch" else 1, s.groups, 0, 1e-5, 0.1,
                            L.F32 if s.compute_dtype == "f32" else L.BF16, 0)

    def _workspaces(self, desc, device):
        key = (desc.B, desc.H, desc.W, str(device))
        hit = self._scratch.get(key)
        if hit is None:
            sb, cb = ctypes.c_size_t(), ctypes.c_size_t()
            L.call("p4c_halfunet_workspace_bytes", ctypes.byref(desc), ctypes.byref(sb), ctypes.byref(cb))
            hit = (sb.value, torch.empty(cb.value, dtype=torch.uint8, device=device))
            self._scratch = {key: hit}  # one shape at a time
        return hit

    # ---------------------------------------------------------------- nn.Module API
    def padding_for(self, H: int, W: int):
        """(top, bottom, left, right) zero padding that takes (H, W) to the next multiple of 16 (four 2x2 poolings), centred as
        mfai's AutoPaddingModel does (extra row / column at the end); all zeros when the grid already fits."""
        dh, dw = (-H) % 16, (-W) % 16
        return dh // 2, dh - dh // 2, dw // 2, dw - dw // 2

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: (B,H,W,in_channels) -> (B,H,W,out_channels)."""
        L.require_cuda(x)
        H, W = x.shape[1], x.shape[2]
        top, bottom, left, right = self.padding_for(H, W)
        if top or bottom or left or right:
            if not self._settings.autopad_enabled:
                raise L.P4CError(f"HalfUNetMI355X: grid {H}x{W} must be a multiple of 16 in both dimensions "
                                 "(or set autopad_enabled, config/CLI/model/halfunet.yaml:26)")
            # pad -> plan -> crop; both are torch views / copies, differentiable (the padded border's outputs are dropped)
            y = self.forward(torch.nn.functional.pad(x, (0, 0, left, right, top, bottom)))
            return y[:, top: top + H, left: left + W, :]
        if self.module_path:
            return self._forward_modules(x)
        if x.shape[-1] == self.in_channels and self.cin_pad != self.in_channels:
            x = torch.nn.functional.pad(x, (0, self.cin_pad - self.in_channels))
        elif x.shape[-1] != self.cin_pad:
            raise L.P4CError(f"HalfUNetMI355X: expected {self.in_channels} (or padded {self.cin_pad}) channels, got {x.shape[-1]}")
        out_dtype = x.dtype
        x = x.contiguous().to(self.act_dtype)
        self._running = self._running_stats(x.device)
        training = self.training
        if training and self._settings.norm == "batch":
            torch._foreach_add_([nb.num_batches_tracked for nb in self._norms], 1)
        y = _HalfUNetFn.apply(x, self, training, *self._ordered_params())
        y = y[..., : self.out_channels]
        return y if y.dtype == out_dtype or not out_dtype.is_floating_point else y.to(out_dtype)

    # ---------------------------------------------------------------- native rollout (one autograd node)
    def native_rollout(self, lm, batch, std, mean, border_flat, interior_flat, force_border):
        """
        Whole training/validation rollout of AutoRegressiveLightning._common_step (lightning.py:565-662) as ONE
        autograd node: per AR step K1 (build x, padded layout) -> HalfUNet plan -> fused state update + border
        forcing + weighted loss, and the matching reverse sweep (BPTT) enqueued back to back from Python with no
        autograd bookkeeping in between.  Returns the (B,T,*S,F) prediction, with ``fused_loss`` (B,T) attached
        when the configured loss is a single WeightedLoss.  Returns None when the configuration is not covered
        (the caller then takes the generic per-op path).
        """
        from .losses import WeightedLoss

        if batch.num_input_steps != 1 or batch.inputs.tensor.dim() != 5:
            return None
        if any(self.padding_for(batch.inputs.tensor.shape[2], batch.inputs.tensor.shape[3])):
            return None   # auto-padded grids take the generic per-step path (forward pads and crops around the plan)
        members = getattr(lm.loss, "losses", [])
        if len(members) != 1 or not isinstance(members[0][0], WeightedLoss) or not members[0][0].fused_capable:
            return None
        wl, wl_weight = members[0]
        weights = wl.weights(tuple(batch.outputs.feature_names), batch.inputs.tensor.device)
        mode = L.MASK_FROM_NAN if lm.mask_on_nan else L.MASK_NONE
        self._running = self._running_stats(batch.inputs.tensor.device)
        training = self.training
        if training and self._settings.norm == "batch":
            torch._foreach_add_([nb.num_batches_tracked for nb in self._norms], batch.num_pred_steps)
        pred, loss = _NativeRolloutFn.apply(
            self, lm, batch.inputs.tensor, batch.forcing.tensor, batch.outputs.tensor,
            lm.grid_static_features[: batch.batch_size], std, mean, border_flat, interior_flat, bool(force_border),
            weights, float(wl.num_interior), wl.kind, mode, training, torch.is_grad_enabled(), *self._ordered_params())
        pred.fused_loss = loss * wl_weight
        return pred

    def roofline(self, ktimes, B, H, W):
        """bench.py: achieved rate of the dominant kernel (conv 3x3 64->64 at full resolution) from the per-launch
        HIP-event timings collected by p4c_prof_*.  The roofline figure uses the launches of the FORWARD plan, where the
        kernel has the device to itself; the same kernel's data-gradient launches and the weight-gradient kernel run
        beside each other in the backward plan (side stream) and are reported separately."""
        ms, n, units = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
        L.lib().p4c_prof_collect(L.PROF_CONV3X3_C64, B * H * W, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(units))
        if n.value == 0:
            return None
        flops = 2.0 * 9 * 64 * 64 * units.value  # algorithmic: 2*taps*Cin*Cout per output pixel
        tflops = flops / (ms.value * 1e-3) / 1e12
        bf = self._settings.compute_dtype == "bf16"
        peak = 2500.0 if bf else 157.3  # dense MFMA peaks, MI355X_MICROARCH.md
        if bf:
            # at the bf16 MFMA rate the kernel is HBM-bound: algorithmic bytes = read 64 ch + write 64 ch per pixel
            esz = 2 if self.act_dtype == torch.bfloat16 else 4
            gbs = 2.0 * 64 * esz * units.value / (ms.value * 1e-3) / 1e9
            kind = L.lib().p4c_conv_kernel_kind(L.BF16, L.BF16 if esz == 2 else L.F32, 64, 3, B, H, W)
            kname = {2: "conv3x3_bf16_rows_kernel", 1: "conv3x3_bf16_ring_kernel"}.get(kind, "conv_fwd_bf16_ws_kernel<f32,64,3>")
            traffic, traffic_source = None, None
            if esz == 2 and (B, H, W) == (2, 512, 512):
                # HBM bytes per launch of this launch shape from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE with the
                # gfx950 corrections): a committed figure, reported only when it was measured on this tree's kernel sources
                traffic, traffic_source = L.committed_traffic(("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"),
                                                              (kname, "hbm_bytes_per_launch"))
            return {"bound": "hbm", "kernel": kname + " (3x3 conv 64->64, forward-plan launches at full resolution)",
                    "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": traffic,
                    "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": 2.0 * 64 * esz * B * H * W,
                    "avg_launch_ms": ms.value / n.value, "launches": n.value, "mfma_tflops": tflops,
                    # context, a reference value of an earlier micro-benchmark (profiles/r01_mfma_rate_microbench.txt): on random
                    # operands the matrix pipe sustains 22.7 ns per v_mfma_f32_32x32x16_bf16 per SIMD (power-limited clock)
                    "mfma_sustained_tflops_reference": 1478.0, "mfma_frac_of_sustained_reference": tflops / 1478.0}
        return {"bound": "mfma", "kernel": "conv_fwd_f32_kernel<64,3,4> (3x3 conv 64->64, forward-plan launches at full resolution)",
                "achieved": tflops, "peak": peak, "unit": "TFLOP/s", "frac": tflops / peak, "traffic": None,
                "avg_launch_ms": ms.value / n.value, "launches": n.value, "flops_per_launch": flops / n.value}

    def step_algorithmic_bytes(self, B, H, W, F, Fs, Ff, T):
        """bench.py, `roofline.step`: HBM bytes one training step (T-step rollout + loss + BPTT + AdamW) has to move in the plan's own
        formulation -- every launch's compulsory reads and writes, each map counted once per launch that needs it, nothing for halos,
        per-workgroup partials, statistics or parameters (< 1 % together).  Table (also in DESIGN.md section 6): M = one 64-channel
        map at full resolution in the activation type, level k of the U-Net moves M / 4^k.
        Returns (total bytes, {family: bytes})."""
        esz = 2 if self.act_dtype == torch.bfloat16 else 4
        n0 = float(B * H * W)
        M = 64.0 * esz * n0
        q = [0.25 ** k for k in range(5)]
        cpad = self.cin_pad
        fwd = {
            "conv 3x3 first (x -> 64)": cpad * esz * n0 + M,
            "conv 3x3 64->64 full resolution (enc1.2, dec.1, dec.2)": 3 * 2 * M,
            "conv 3x3 64->64 coarse levels (8 launches)": sum(2 * 2 * M * q[k] for k in range(1, 5)),
            "max-pool (4)": sum(M * q[k] + M * q[k + 1] for k in range(4)),
            "up-sample + sum of the five levels": M * sum(q) + M,
            # (bf16 flavour: the 1x1 output convolution runs inside the AR step's kernel -- p4c_out_conv_update_loss_fwd --, its
            # output is neither written nor read back)
            "conv 1x1 output": M if esz == 2 else 2 * M,
        }
        # backward of one AR step.  Per 3x3 block at level k: pass 1 of the normalisation backward (dA + y; where its producer
        # takes it -- the 1x1 data gradient for block 11, enc_out_bwd for blocks 1, 3, 5, 7, 9 -- only y is read again), the data
        # gradient (dA + y in, dX out; pass 2 of the normalisation backward is formed on the way in), the weight gradient
        # (x + dA + y in).  The first convolution's data gradient exists only where the previous state needs one (AR steps > 0).
        # bf16 flavour: the 1x1 convolution's data gradient, weight gradient and block 11's pass 1 are ONE kernel (out_conv_bwd.hip):
        # dy + y in, dA out.
        lev = [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 0, 0]
        fused_p1 = {11, 1, 3, 5, 7, 9}
        nbwd = sum((0 if (i == 11 and esz == 2) else 1 if i in fused_p1 else 2) * M * q[lev[i]] for i in range(12))
        wgrad = sum((2 * M + (cpad * esz * n0 if i == 0 else M)) * q[lev[i]] for i in range(12))
        dgrad_rest = sum(3 * M * q[lev[i]] for i in range(1, 12))
        bwd = {
            "conv 1x1: data gradient (dy in, dA out) + weight gradient (y + dy)": 3 * M if esz == 2 else 2 * M + 2 * M,
            "normalisation backward, pass 1 (12 blocks)": nbwd,
            "conv 3x3 data gradients (blocks 1..11)": dgrad_rest,
            "conv 3x3 weight gradients (12 blocks)": wgrad,
            "up-sampling adjoints, x pass (dS in, four half-width maps out)": M + M * (0.5 + 0.25 + 0.125 + 0.0625),
            "encoder outputs' gradients (y pass of the adjoint + pool adjoint + ReLU mask: enc_out_bwd, 5 levels)":
                sum((M * (0.5 ** k if k else 1.0)) + (M * q[k + 1] if k < 4 else 0.0) + 2 * M * q[k] for k in range(5)),
        }
        dgrad0 = 3 * M   # AR steps 1 .. T-1
        f4 = 4.0 * n0
        rollout = {
            "build_x (AR step 0)": f4 * (F + Fs + Ff) + cpad * esz * n0,
            "state update + loss forward (prev, target, y in; new state, saved loss gradient out)":
                T * (f4 * 3 * F + (1 if esz == 2 else 2) * esz * F * n0),
            "next network input emitted by the update (AR steps 0 .. T-2)": (T - 1) * (f4 * (Fs + Ff) + cpad * esz * n0),
            "state update + loss backward": T * (esz * F * n0 + M) + (T - 1) * (2 * f4 * F + esz * F * n0),
        }
        table = {f"forward: {k}": T * v for k, v in fwd.items()}
        table.update({f"backward: {k}": T * v for k, v in bwd.items()})
        table["backward: conv 3x3 first, data gradient (AR steps > 0)"] = (T - 1) * dgrad0
        table.update({f"rollout: {k}": v for k, v in rollout.items()})
        npar = float(sum(p.numel() for p in self.parameters()))
        table["optimizer (AdamW: p, g, m, v in; p, m, v out) + gradient clear"] = 4.0 * npar * 8
        return sum(table.values()), table

    def launch_times(self, B, H, W):
        """bench.py, one extra un-timed step with every tagged launch bracketed: average duration of the roofline kernel's
        data-gradient launches and of the weight-gradient kernel at full resolution (they overlap each other in the backward
        plan -- main and side stream -- so they are reported next to, not inside, the roofline figure)."""
        out = {}
        for key, tag in (("datagrad_avg_launch_ms_overlapped", L.PROF_CONV3X3_C64_BWD), ("wgrad_avg_launch_ms_overlapped", L.PROF_WGRAD3X3_C64)):
            ms, n, units = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
            L.lib().p4c_prof_collect(tag, B * H * W, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(units))
            out[key] = (ms.value / n.value) if n.value else None
        return out


class _NativeRolloutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, lm, inputs, forcing, outputs, statics, std, mean, border_flat, interior_flat, force_border,
                weights, num_interior, kind, mask_mode, training, keep_saved, *params):
        L.require_cuda(inputs, forcing, outputs)
        dev = inputs.device
        B, T = outputs.shape[0], outputs.shape[1]
        H, W, F = inputs.shape[2], inputs.shape[3], inputs.shape[4]
        N = H * W
        Fs, Ff = statics.shape[-1], forcing.shape[-1]
        inputs, forcing, outputs = inputs.float().contiguous(), forcing.float().contiguous(), outputs.float().contiguous()
        st = statics.float()
        sbs = 0 if st.stride(0) == 0 else N * Fs
        st = st[0].contiguous() if sbs == 0 else st.contiguous()
        desc = model._desc(B, H, W)
        flat = model._flat_params()
        saved_bytes, scratch = model._workspaces(desc, dev)
        cpad = model.cin_pad
        mask_on_nan = int(mask_mode == L.MASK_FROM_NAN)
        count = None
        if mask_on_nan:
            count = torch.empty(1, dtype=torch.int32, device=dev)
            L.call("p4c_mask_all_zero_count", L.ptr(outputs), mask_mode, T * N * F, N * F, B, T, N, F, L.ptr(count), L.stream(dev))
        # state buffer: slot 0 = input state, slot i+1 = new state of AR step i (= prediction[:, i])
        # (slot 0 is never written: AR step 0 reads the batch's input state where it lies -- every kernel takes the previous
        # state's batch stride separately -- which saves a 126 MB copy per rollout at the benchmark size)
        states = torch.empty(B, T + 1, H, W, F, dtype=torch.float32, device=dev)
        sbs_state = (T + 1) * N * F
        sbs_input = inputs.shape[1] * N * F
        loss = torch.empty(B, T, dtype=torch.float32, device=dev)
        ws = torch.empty(L.lib().p4c_loss_workspace_bytes(B, 1, N, 1) // 4, dtype=torch.float32, device=dev)
        xs, saveds = [], []
        adt, acode = model.act_dtype, L.dtype_code(model.act_dtype)
        y = torch.empty(B, H, W, NF, dtype=adt, device=dev)
        stream = L.stream(dev)
        saved = None
        # the parameters are fixed for the whole rollout: re-lay / round the weights once, not once per AR step
        L.call("p4c_halfunet_prepare_weights", ctypes.byref(desc), L.ptr(flat), L.ptr(scratch), stream)
        desc.weights_prepared = 1
        # "feed next step" fused: the update kernel of step i also writes step i+1's network input (new state | statics |
        # next forcing | padding), so only step 0 runs p4c_build_x.  (The NaN-mask input channel needs p4c_build_x.)
        lanes = 1
        while lanes < F // 4:
            lanes *= 2  # lanes per grid point of the 16-byte update kernel; each also owns one tail quad of x_next
        v4_next = ((not mask_on_nan) and F % 4 == 0 and F <= 64 and Fs % 4 == 0 and cpad % 4 == 0
                   and cpad // 4 - F // 4 <= lanes)
        # any other feature count (the shipped Titan configuration has 21 features): the flat kernels of csrc/losses.hip -- (N, F)
        # arrays streamed flat, the network's row tensors through LDS tiles -- take the same fused step
        esz = 2 if adt == torch.bfloat16 else 4
        flat_next = ((not mask_on_nan) and F <= 64 and (N * F) % 4 == 0 and (cpad * esz) % 16 == 0 and cpad <= 256
                     and L.diag_switch("P4C_NO_FLAT_STEP") != "1")
        fuse_next = v4_next or flat_next
        x_next = None
        # bf16 flavour, training: the update kernel also saves the loss gradient of every element as bf16 rows and the backward reads
        # those instead of the new state and the target (480 -> 120 bytes per grid point; P4C_SAVE_LOSS_GRAD=0: recompute)
        save_lg = (keep_saved and adt == torch.bfloat16 and fuse_next and mask_mode == L.MASK_NONE
                   and L.diag_switch("P4C_SAVE_LOSS_GRAD") != "0")
        lgrads = torch.empty(T, B, N, F, dtype=torch.bfloat16, device=dev) if save_lg else None
        # bf16 flavour: the network's 1x1 output convolution runs INSIDE the AR step's kernel (p4c_out_conv_update_loss_fwd: y is
        # never written and read back, one launch less per AR step; same new state bit for bit; P4C_FUSED_TAIL=0: the two-kernel route)
        # (feature counts off the 16-byte grid: the flat kernel takes the convolution as its front end)
        fused_tail = (adt == torch.bfloat16 and (v4_next or (flat_next and F % 4 != 0)) and mask_mode == L.MASK_NONE
                      and model.out_channels >= F and L.diag_switch("P4C_FUSED_TAIL") != "0")
        desc_fwd = desc
        if fused_tail:
            desc_fwd = HalfUNetDesc.from_buffer_copy(desc)
            desc_fwd.skip_out_conv = 1
        for i in range(T):
            prev, sbs_prev = (inputs[:, 0], sbs_input) if i == 0 else (states[:, i], sbs_state)
            if x_next is not None:
                x = x_next
            else:
                x = torch.empty(B, H, W, cpad, dtype=adt, device=dev)
                L.call("p4c_build_x", L.ptr(prev), sbs_prev, N * F, L.ptr(st), sbs, L.ptr(forcing[:, i]), T * N * Ff,
                       L.ptr(x), acode, cpad, B, 1, N, F, Fs, Ff, mask_on_nan, 0, stream)
            if saved is None or keep_saved:
                saved = torch.empty(saved_bytes, dtype=torch.uint8, device=dev)
            L.call("p4c_halfunet_forward", ctypes.byref(desc_fwd), L.ptr(x), L.ptr(flat), L.ptr(model._running),
                   None if fused_tail else L.ptr(y), L.ptr(saved), L.ptr(scratch), int(training), stream)
            if fused_tail:
                last = i + 1 == T
                x_next = None if last else torch.empty(B, H, W, cpad, dtype=adt, device=dev)
                ta, tsc, tsh, tw = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
                L.call("p4c_halfunet_tail", ctypes.byref(desc_fwd), L.ptr(flat), L.ptr(saved), ctypes.byref(ta), ctypes.byref(tsc),
                       ctypes.byref(tsh), ctypes.byref(tw))
                L.call("p4c_out_conv_update_loss_fwd", ta, tsc, tsh, tw, model.out_channels,
                       L.ptr(prev), sbs_prev, L.ptr(outputs[:, i]), T * N * F, L.ptr(std), L.ptr(mean),
                       L.ptr(border_flat if force_border else None), L.ptr(interior_flat), L.ptr(states[:, i + 1]), sbs_state,
                       L.ptr(weights), num_interior, L.ptr(count), kind, L.ptr(loss[:, i]), T, L.ptr(ws), B, N, F, 1.0,
                       L.ptr(x_next), cpad, L.ptr(st), sbs, Fs, L.ptr(None if last else forcing[:, i + 1]), T * N * Ff, Ff,
                       L.ptr(lgrads[i]) if save_lg else None, N * F, stream)
                if keep_saved:
                    xs.append(x)
                    saveds.append(saved)
                continue
            step_args = (L.ptr(prev), sbs_prev, L.ptr(y), acode, NF, L.ptr(outputs[:, i]),
                         T * N * F, L.ptr(std), L.ptr(mean), L.ptr(border_flat if force_border else None), L.ptr(interior_flat),
                         L.ptr(states[:, i + 1]), sbs_state, L.ptr(weights), num_interior, L.ptr(count), kind, mask_mode,
                         L.ptr(loss[:, i]), T, L.ptr(ws), B, N, F, 1.0)
            x_next = None
            if save_lg:
                last = i + 1 == T
                x_next = None if last else torch.empty(B, H, W, cpad, dtype=adt, device=dev)
                L.call("p4c_ar_update_loss_fwd_next_saved", *step_args, L.ptr(x_next), cpad, L.ptr(st), sbs, Fs,
                       L.ptr(None if last else forcing[:, i + 1]), T * N * Ff, Ff, L.ptr(lgrads[i]), N * F, stream)
            elif fuse_next and i + 1 < T:
                x_next = torch.empty(B, H, W, cpad, dtype=adt, device=dev)
                L.call("p4c_ar_update_loss_fwd_next", *step_args, L.ptr(x_next), cpad, L.ptr(st), sbs, Fs,
                       L.ptr(forcing[:, i + 1]), T * N * Ff, Ff, stream)
            else:
                L.call("p4c_ar_update_loss_fwd", *step_args, stream)
            if keep_saved:
                xs.append(x)
                saveds.append(saved)
        ctx.model, ctx.desc, ctx.training = model, desc, training
        ctx.meta = (B, T, H, W, F, force_border, num_interior, kind, mask_mode)
        ctx.tensors = (states, outputs, std, interior_flat, weights, count, xs, saveds)
        ctx.lgrads = lgrads
        ctx.set_materialize_grads(False)
        pred = states[:, 1:]
        return pred, loss

    @staticmethod
    def backward(ctx, g_pred, g_loss):
        model, desc = ctx.model, ctx.desc
        B, T, H, W, F, force_border, num_interior, kind, mask_mode = ctx.meta
        states, outputs, std, interior_flat, weights, count, xs, saveds = ctx.tensors
        dev = states.device
        N = H * W
        stream = L.stream(dev)
        flat = model._flat_params()
        _, scratch = model._workspaces(desc, dev)
        target = model._flat_grad_target()
        gflat = target if target is not None else torch.zeros_like(flat)
        adt, acode = model.act_dtype, L.dtype_code(model.act_dtype)
        dy = torch.empty(B, H, W, NF, dtype=adt, device=dev)
        dx = torch.empty(B, H, W, NF, dtype=adt, device=dev)
        dprev = torch.empty(B, H, W, F, dtype=torch.float32, device=dev)
        gl = g_loss.contiguous().float() if g_loss is not None else None
        sbs_state = (T + 1) * N * F
        # the scratch workspace may have served another call since forward: prepare again (one launch per sweep)
        L.call("p4c_halfunet_prepare_weights", ctypes.byref(desc), L.ptr(flat), L.ptr(scratch), stream)
        desc0 = HalfUNetDesc.from_buffer_copy(desc)
        desc0.dx_channels = 0  # the input state of step 0 is data: no gradient needed, skip that conv
        have_next = False
        # Weight gradients run on the library's side stream.  Their join is deferred over the whole reverse sweep: the
        # full-resolution ones left over at the end of AR step i's backward then run beside the HBM-bound head of step i-1's
        # chain instead of alone (P4C_DEFER_JOIN=0: join after every step, the round-2 behaviour).  xs / saveds stay referenced
        # until the join below -- the side stream reads them.
        defer = T > 1 and L.diag_switch("P4C_DEFER_JOIN") != "0"
        if defer:
            L.call("p4c_side_stream_defer", 1)
        ok = False
        try:
            out = _NativeRolloutFn._sweep(ctx, g_pred, gl, dy, dx, dprev, gflat, flat, scratch, desc, desc0, stream, target, defer)
            ok = True
            return out
        finally:
            if defer:
                L.call("p4c_side_stream_defer", 0)
                if not ok:
                    # the sweep raised between two calls: weight gradients already on the side stream still read xs / saveds / dy,
                    # which the unwinding is about to release to the allocator -- make the caller's stream (the one the allocator
                    # orders re-use on) wait for them first.  (The join itself must not mask the original error.)
                    try:
                        L.call("p4c_side_stream_join", stream)
                    except Exception:  # noqa: BLE001
                        pass

    @staticmethod
    def _sweep(ctx, g_pred, gl, dy, dx, dprev, gflat, flat, scratch, desc, desc0, stream, target, defer):
        model = ctx.model
        B, T, H, W, F, force_border, num_interior, kind, mask_mode = ctx.meta
        states, outputs, std, interior_flat, weights, count, xs, saveds = ctx.tensors
        N = H * W
        sbs_state = (T + 1) * N * F
        adt, acode = model.act_dtype, L.dtype_code(model.act_dtype)
        have_next = False
        for i in range(T - 1, -1, -1):
            g_next = dprev if have_next else None
            if g_pred is not None:  # somebody differentiates through the prediction itself: add its slice
                if g_next is None:
                    dprev.copy_(g_pred[:, i])
                else:
                    dprev.add_(g_pred[:, i])
                g_next = dprev
            if ctx.lgrads is not None:
                L.call("p4c_ar_update_loss_bwd_saved", L.ptr(g_next), N * F, L.ptr(dx if have_next else None), acode, NF,
                       L.ptr(gl[:, i]) if gl is not None else None, T, L.ptr(ctx.lgrads[i]), N * F, L.ptr(std), L.ptr(interior_flat),
                       int(force_border), L.ptr(weights), num_interior, L.ptr(count), kind, mask_mode, L.ptr(dy), acode, NF,
                       L.ptr(dprev) if i > 0 else None, N * F, B, N, F, 1.0, stream)
            else:
                L.call("p4c_ar_update_loss_bwd", L.ptr(g_next), N * F, L.ptr(dx if have_next else None), acode, NF,
                       L.ptr(gl[:, i]) if gl is not None else None, T, L.ptr(states[:, i + 1]), sbs_state, L.ptr(outputs[:, i]),
                       T * N * F, L.ptr(std), L.ptr(interior_flat), int(force_border), L.ptr(weights), num_interior,
                       L.ptr(count), kind, mask_mode, L.ptr(dy), acode, NF, L.ptr(dprev) if i > 0 else None, N * F, B, N, F,
                       1.0, stream)
            d = desc if i > 0 else desc0
            L.call("p4c_halfunet_backward", ctypes.byref(d), L.ptr(xs[i]), L.ptr(flat), L.ptr(dy),
                   L.ptr(dx) if i > 0 else None, L.ptr(gflat), L.ptr(saveds[i]), L.ptr(scratch), int(ctx.training), stream)
            have_next = True
            if not defer:
                xs[i] = None
                saveds[i] = None  # release the step's activations as soon as its backward is enqueued
        if defer:
            L.call("p4c_side_stream_join", stream)
            for i in range(T):
                xs[i] = None
                saveds[i] = None
        if target is not None:  # already accumulated into param.grad
            return (None,) * (17 + len(model._param_slices))
        grads = tuple(gflat[o : o + n].view(s) for (o, n, s) in model._param_slices)
        return (None,) * 17 + grads
