"""
``AutoRegressiveLightning`` on MI355X: same constructor, attributes and step methods as
``py4cast/lightning.py:147-1188`` of the reference, with the autoregressive rollout
(``_common_step`` :495-676, ``_next_x`` :711-767) and the training loss evaluated by the HIP
kernels of ``libpy4cast_hip.so``.

Drop-in boundary: ``bin/main.py`` of this repo passes this class to ``Py4castLightningCLI``
instead of the reference's (one-line difference from the reference's ``bin/main.py:12``).
When ``lightning`` is not installed the class derives from ``torch.nn.Module`` and is driven
by ``py4cast_amd.trainer.Trainer`` (a minimal fit loop with the same hook names).

What changed relative to the reference, and why:
* diff-stats (std/mean) and loss weights are uploaded once per device instead of being
  re-stacked on CPU and copied H2D at every AR step (lightning.py:696-709);
* ``_next_x`` + layout change, the residual update + border blend, and the loss are each ONE
  kernel instead of ~12 elementwise/copy kernels per step;
* ``get_mask_on_nan`` returns a marker, not a materialised mask (lightning.py:797 allocates a
  full ``ones_like`` per step);
* models exposing ``native_rollout`` (the HIP HalfUNet) run the whole rollout + loss + BPTT
  as one autograd node enqueued from C++.
"""

import math
import os
from copy import deepcopy
from functools import cached_property
from pathlib import Path
from typing import Dict, List, Literal, Optional, Tuple, Union

import torch

from . import _lib as L
from . import ops
from .base import ItemBatch, ModelType, expand_to_batch, features_last_to_second, features_second_to_last
from .losses import CombinedLoss, NanMask, OnesMask, ScaledLoss, WeightedLoss
from .models import build_model_from_settings, get_model_kls_and_settings
from .models import registry as model_registry
from .namedtensor import NamedTensor
from .optim import FlatAdamW

try:  # pragma: no cover - lightning absent in the build image
    from lightning import LightningModule as _Base  # type: ignore

    HAVE_LIGHTNING = True
except Exception:
    HAVE_LIGHTNING = False

    class _Base(torch.nn.Module):
        """Just enough of LightningModule for the hot path when lightning is absent."""

        def __init__(self, *args, **kwargs):
            super().__init__()
            self.trainer = None
            self._logged = {}

        def save_hyperparameters(self, *args, **kwargs):
            import inspect

            frame = inspect.currentframe().f_back
            names = frame.f_code.co_varnames[1 : frame.f_code.co_argcount]
            self.hparams = _AttrDict({n: frame.f_locals[n] for n in names if n in frame.f_locals})

        def log(self, name, value, **kwargs):
            self._logged[name] = value

        def log_dict(self, d, **kwargs):
            self._logged.update(d)

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")


class _AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


str_to_dtype = {  # py4cast/utils.py:104-109
    "bf16-true": torch.bfloat16,
    "16-true": torch.float16,
    "32-true": torch.float32,
    "64-true": torch.float64,
}


def rank_zero_init(model_kls, model_settings, statics):
    """lightning.py:141-144 (rank_zero_only when lightning is present)."""
    if hasattr(model_kls, "rank_zero_setup"):
        distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        rank = torch.distributed.get_rank() if distributed else 0
        if rank == 0:
            model_kls.rank_zero_setup(model_settings, statics.meshgrid)
        if distributed:
            # the other ranks construct the model right after this call and read what rank 0 wrote (e.g. the mesh graph file)
            torch.distributed.barrier()


def cosine_with_min_lr_lambda(num_warmup_steps: int, num_training_steps: int, min_lr_rate: float, num_cycles: float = 0.5):
    """transformers.get_cosine_with_min_lr_schedule_with_warmup (used at lightning.py:453-458)."""

    def fn(step: int) -> float:
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        factor = 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress))
        return max(0.0, factor * (1 - min_lr_rate) + min_lr_rate)

    return fn


class _LazyMaskedTarget(NamedTensor):
    """``target_masked`` of lightning.py:792-796 (a clone with NaN -> 0) that costs nothing until somebody reads ``.tensor``:
    the loss kernels take the RAW target through the NanMask marker and replace NaNs in registers, so on the hot path the
    NaN-free copy (one full pass over the target per step in the reference) is never made; observers that do read it get the
    reference's tensor."""

    def __init__(self, raw: NamedTensor):   # metadata copied, no tensor work
        self.__dict__.update({k: v for k, v in raw.__dict__.items() if k != "tensor"})
        self.names, self.feature_names = list(raw.names), list(raw.feature_names)
        self._raw, self._clean = raw.tensor, None

    @property
    def tensor(self) -> torch.Tensor:
        if self._clean is None:
            self._clean = torch.nan_to_num(self._raw, nan=0)
        return self._clean

    @tensor.setter
    def tensor(self, value: torch.Tensor):
        self._clean = value


class AutoRegressiveLightning(_Base):
    """Auto-regressive module for predicting meteorological fields (lightning.py:147)."""

    def __init__(
        self,
        settings_init_args: dict,
        dataset_info,
        infer_ds,
        dataset_name: str = "dummy",
        dataset_conf: Optional[Dict] = None,
        num_input_steps: int = 1,
        num_pred_steps_train: int = 1,
        num_pred_steps_val_test: int = 1,
        batch_size: int = 2,
        model_name: str = "HalfUNet",
        losses: List[dict] = [{"class": "WeightedLoss", "params": {"loss": "MSELoss", "reduction": "none"}}],
        num_inter_steps: int = 1,
        num_samples_to_plot: int = 1,
        training_strategy: Literal["diff_ar", "scaled_ar", "downscaling_only"] = "diff_ar",
        channels_last: bool = False,
        io_conf: Optional[Path] = None,
        mask_ratio: float = 0,
        mask_on_nan: bool = False,
        learning_rate: float = 1e-4,
        min_learning_rate: float = 1e-6,
        num_warmup_steps: int = 0,
        betas: tuple = (0.9, 0.999),
        *args,
        **kwargs,
    ):
        super().__init__(*args, **kwargs)
        self.infer_ds = infer_ds
        self.settings_init_args = settings_init_args
        self.dataset_name = dataset_name
        self.dataset_conf = dataset_conf
        self.dataset_info = dataset_info
        self.batch_size = batch_size
        self.model_name = model_name
        self.num_input_steps = num_input_steps
        self.num_pred_steps_train = num_pred_steps_train
        self.num_pred_steps_val_test = num_pred_steps_val_test
        self.num_inter_steps = num_inter_steps
        self.num_samples_to_plot = num_samples_to_plot
        self.training_strategy = training_strategy
        self.channels_last = channels_last
        self.io_conf = io_conf
        self.mask_ratio = mask_ratio
        self.mask_on_nan = mask_on_nan
        self.learning_rate = learning_rate
        self.min_learning_rate = min_learning_rate
        self.num_warmup_steps = num_warmup_steps
        self.betas = betas

        if self.training_strategy == "downscaling_only":
            print("WARNING : You are using downscaling_only mode: this is experimental.")
        if self.num_inter_steps > 1 and self.num_input_steps > 1:  # lightning.py:213-217
            raise AttributeError(
                "It is not possible to have multiple input steps when num_inter_steps > 1."
                f"Get num_input_steps :{self.num_input_steps} and num_inter_steps: {self.num_inter_steps}"
            )
        ALLOWED_STRATEGIES = ("diff_ar", "scaled_ar", "downscaling_only")
        if self.training_strategy not in ALLOWED_STRATEGIES:  # lightning.py:218-222
            raise AttributeError(
                f"Unknown strategy {self.training_strategy}, allowed strategies are {ALLOWED_STRATEGIES}"
            )

        self.save_hyperparameters()
        self.hparams["dataset_info"] = dataset_info
        self.hparams["infer_ds"] = infer_ds

        statics = deepcopy(dataset_info.statics)  # lightning.py:232
        self.diff_stats = dataset_info.diff_stats
        self.stats = dataset_info.stats
        self.grid_shape = statics.grid_shape
        self.plotted_examples = 0
        self.spatial_loss_maps = []
        self.training_step_losses = []
        self.validation_step_losses = []

        num_grid_static_features = statics.grid_statics.dim_size("features")
        ds = self.training_strategy == "downscaling_only"
        num_input_features = (  # lightning.py:256-261
            num_input_steps * dataset_info.weather_dim * (1 - ds)
            + num_grid_static_features
            + dataset_info.forcing_dim
            + self.mask_on_nan
        )
        num_output_features = dataset_info.weather_dim

        model_kls, model_settings = get_model_kls_and_settings(model_name, self.settings_init_args)
        rank_zero_init(model_kls, model_settings, statics)
        self.model, model_settings = build_model_from_settings(
            model_name, num_input_features, num_output_features, self.settings_init_args, statics.grid_shape
        )
        if channels_last:
            self.model = self.model.to(memory_format=torch.channels_last)

        if self.model.model_type == ModelType.GRAPH:  # lightning.py:285-289
            statics.grid_statics.flatten_("ngrid", 0, 1)
            statics.border_mask = statics.border_mask.flatten(0, 1)
            statics.interior_mask = statics.interior_mask.flatten(0, 1)

        statics.register_buffers(self)  # border_mask, interior_mask (H,W,1) / (N,1)
        self.num_spatial_dims = statics.grid_statics.num_spatial_dims
        self.register_buffer(
            "grid_static_features", expand_to_batch(statics.grid_statics.tensor, batch_size), persistent=False
        )

        self.loss = CombinedLoss(losses)
        self.loss.prepare(self, statics.interior_mask, dataset_info)
        self._dev_cache = {}

    # ------------------------------------------------------------------ checkpoint keys (lightning.py:338-354)
    def on_save_checkpoint(self, checkpoint):
        checkpoint["input_feature_names"] = self.input_feature_names
        checkpoint["output_feature_names"] = self.output_feature_names
        checkpoint["output_dim_names"] = self.output_dim_names
        checkpoint["output_dtype"] = self.output_dtype

    def on_load_checkpoint(self, checkpoint):
        self.input_feature_names = checkpoint["input_feature_names"]
        self.output_feature_names = checkpoint["output_feature_names"]
        self.output_dim_names = checkpoint["output_dim_names"]
        self.output_dtype = checkpoint["output_dtype"]

    @property
    def logging_enabled(self) -> bool:
        tr = getattr(self, "trainer", None)
        return bool(tr is not None and getattr(tr, "logger", None) is not None and tr.logger.log_dir is not None)

    @property
    def dtype(self):
        """torch dtype for the precision requested from the trainer (lightning.py:363-368)."""
        tr = getattr(self, "trainer", None)
        precision = getattr(tr, "precision", "32-true") if tr is not None else "32-true"
        return str_to_dtype[precision]

    @cached_property
    def interior_2d(self) -> torch.Tensor:
        if self.num_spatial_dims == 1:
            return self.interior_mask.reshape(self.grid_shape[0], -1, self.interior_mask.shape[-1])
        return self.interior_mask

    def configure_optimizers(self):
        """AdamW + cosine-with-min-lr warmup schedule stepped per optimizer step (lightning.py:442-467)."""
        # a torch.optim.AdamW whose step is one kernel when parameters and gradients are flat (HalfUNetMI355X + FlatDDP)
        optimizer = FlatAdamW(self.parameters(), lr=self.hparams.learning_rate, betas=self.hparams.betas)
        total = getattr(self.trainer, "estimated_stepping_batches", 1000) if self.trainer is not None else 1000
        scheduler = torch.optim.lr_scheduler.LambdaLR(
            optimizer,
            cosine_with_min_lr_lambda(
                self.hparams.num_warmup_steps, total, self.hparams.min_learning_rate / self.hparams.learning_rate
            ),
        )
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": scheduler, "interval": "step", "frequency": 1}}

    # ------------------------------------------------------------------ forward
    def forward(self, x: ItemBatch, batch_idx: int) -> NamedTensor:
        return self.common_step(x, batch_idx, phase="inference")[0]

    def common_step(self, batch: ItemBatch, batch_idx: int, phase: str) -> Tuple[NamedTensor, NamedTensor]:
        """
        lightning.py:479-493.  The reference wraps the rollout in torch.autocast; here the
        compute precision is a property of the model kernels (``model.compute_dtype`` for HIP
        models), and the state update / loss always run in fp32 -- the same promotion the
        reference gets from its fp32 std/mean and out-of-autocast loss (:605-610, :816).
        Models without native kernels still get the reference's autocast.
        """
        if getattr(self.model, "is_native_hip", False) or self.dtype == torch.float32:
            return self._common_step(batch, batch_idx, phase)
        with torch.amp.autocast("cuda", dtype=self.dtype):
            return self._common_step(batch, batch_idx, phase)

    def _strategy_params(self) -> Tuple[bool, bool, int]:
        """lightning.py:678-694."""
        force_border = self.training_strategy == "scaled_ar"
        scale_y = self.training_strategy == "scaled_ar"
        if self.training_strategy == "diff_ar" and self.num_inter_steps != 1:
            raise ValueError("Diff AR strategy requires exactly 1 intermediary step.")
        return force_border, scale_y, self.num_inter_steps

    def _step_diffs(self, feature_names: List[str], device: torch.device) -> Tuple[torch.Tensor, torch.Tensor]:
        """lightning.py:696-709, hoisted: built and uploaded once per (feature names, device)."""
        key = ("diff", tuple(feature_names), str(device))
        hit = self._dev_cache.get(key)
        if hit is None:
            hit = (
                self.diff_stats.to_list("std", feature_names).to(device).contiguous(),
                self.diff_stats.to_list("mean", feature_names).to(device).contiguous(),
            )
            self._dev_cache[key] = hit
        return hit

    def _flat_masks(self, device) -> Tuple[torch.Tensor, torch.Tensor]:
        key = ("masks", str(device))
        hit = self._dev_cache.get(key)
        if hit is None:
            hit = (
                self.border_mask.detach().reshape(-1).to(device=device, dtype=torch.float32).contiguous(),
                self.interior_mask.detach().reshape(-1).to(device=device, dtype=torch.float32).contiguous(),
            )
            self._dev_cache[key] = hit
        return hit

    def _record_names(self, batch: ItemBatch, ds: bool):
        """lightning.py:541-558."""
        self.input_feature_names = batch.inputs.feature_names
        self.output_feature_names = batch.outputs.feature_names
        self.output_dim_names = batch.outputs.names
        self.output_dtype = batch.outputs.tensor.dtype
        if ds:
            common = []
            for out_name in self.output_feature_names:
                for i, forcing_name in enumerate(batch.forcing.feature_names):
                    if out_name.split("_")[1:] == forcing_name.split("_")[1:]:
                        common.append(i)
            self.common_features_idx = common

    def _next_x(self, batch: ItemBatch, prev_states: NamedTensor, step_idx: int, c_pad=None, dtype=torch.float32,
                blocks=None):
        """lightning.py:711-767 as one kernel (K1).  Returns (B,*S,C_in[+pad]).  ``blocks``: the block mask of
        ``mask_tensor`` (:580-581), applied by the same kernel."""
        forcing_i = batch.forcing.select_tensor_dim("timestep", step_idx)
        ds = self.training_strategy == "downscaling_only"
        return ops.build_x(
            prev_states.tensor, self.grid_static_features[: batch.batch_size], forcing_i, self.mask_on_nan, ds,
            c_pad=c_pad, dtype=dtype, blocks=blocks,
        )

    def _common_step(self, batch: ItemBatch, batch_idx: int, phase: str) -> Tuple[NamedTensor, NamedTensor]:
        """lightning.py:495-676 on HIP kernels (generic path: any nn.Module model)."""
        force_border, scale_y, num_inter_steps = self._strategy_params()
        self.original_shape = None
        ds = self.training_strategy == "downscaling_only"
        inference = phase == "inference"

        if self.model.model_type == ModelType.GRAPH:  # lightning.py:526-535 (mutates the batch, as the reference)
            self.original_shape = batch.inputs.tensor.shape
            batch.inputs.flatten_("ngrid", *batch.inputs.spatial_dim_idx)
            if not inference:
                batch.outputs.flatten_("ngrid", *batch.outputs.spatial_dim_idx)
            batch.forcing.flatten_("ngrid", *batch.forcing.spatial_dim_idx)

        if batch_idx == 0:
            self._record_names(batch, ds)

        device = batch.inputs.tensor.device
        border_flat, interior_flat = self._flat_masks(device)
        std = mean = None
        if scale_y:
            std, mean = self._step_diffs(
                self.output_feature_names if inference else batch.outputs.feature_names, device
            )

        native = getattr(self.model, "native_rollout", None) if getattr(self, "use_native_rollout", True) else None
        if native is not None and not ds and self.mask_ratio == 0 and num_inter_steps == 1 and not inference:
            prediction = native(self, batch, std, mean, border_flat, interior_flat, force_border)
            if prediction is not None:
                pred_out = NamedTensor.new_like(prediction.type_as(batch.outputs.tensor), batch.outputs)
                pred_out.fused_loss = getattr(prediction, "fused_loss", None)
                return pred_out, batch.outputs

        prev_states = batch.inputs
        prediction_list = []
        T = batch.num_pred_steps
        # (single process, gradients on, several model calls): per-call stand-ins of the parameters, trainer.RolloutParamProxies.
        # By default only while the step is being captured into a HIP graph: swapping the parameters in costs the host ~1.5 ms per
        # model call (SwinUNetR eager: 62.8 -> 67.1 ms per step), nothing in a replay (30.3 -> 29.7); ``use_param_proxies`` forces it.
        proxies = None
        want = getattr(self, "use_param_proxies", None)
        if want is None:
            want = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if (not inference and torch.is_grad_enabled() and T * num_inter_steps > 1 and getattr(self.model, "rollout_param_proxies", False)
                and want and L.diag_switch("P4C_NO_PARAM_PROXIES") != "1"
                and not (torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1)):
            from .trainer import RolloutParamProxies

            proxies = self.__dict__.get("_param_proxies")
            if proxies is None or proxies.model is not self.model:
                proxies = self.__dict__["_param_proxies"] = RolloutParamProxies(self.model)
            proxies.begin()
        call_model = self.model if proxies is None else proxies.call
        # A model may take its input straight from build_x in the layout its first kernels want -- (dtype, channel count padded with
        # zeros), e.g. bf16 rows of 96 channels -- instead of fp32 rows it would cast and pad itself (two passes over the full-resolution
        # input per AR step, and their adjoints); its output may then come back in that dtype too (the state update takes any).
        x_format = {}
        fmt = getattr(self.model, "rollout_input_format", None)
        if fmt is not None and not self.channels_last and not self.model.features_second and getattr(self, "use_rollout_input_format", True):
            x_format = {"dtype": fmt[0], "c_pad": fmt[1]}
        keep_prev = 0.0 if ds else 1.0
        # Any nn.Module model: when the configured loss is a single WeightedLoss, every AR step's state update, border
        # forcing and loss column are ONE kernel (p4c_ar_update_loss_fwd) writing straight into the (B,T,...) prediction
        # buffer -- no separate loss passes over the stacked prediction, no torch.stack copy (lightning.py:599-660,816).
        members = getattr(self.loss, "losses", [])
        fuse = (not inference and not ds and num_inter_steps == 1 and len(members) == 1
                and isinstance(members[0][0], WeightedLoss) and members[0][0].fused_capable and getattr(self, "use_fused_step", True))
        if fuse:
            wl, wl_weight = members[0]
            f_weights = wl.weights(tuple(batch.outputs.feature_names), device)
            f_mode = L.MASK_FROM_NAN if self.mask_on_nan else L.MASK_NONE
            f_count = ops.masked_count(ops.MaskSpec(f_mode), batch.outputs.tensor) if self.mask_on_nan else None
            f_states = torch.empty(batch.outputs.tensor.shape, dtype=torch.float32, device=device)
            f_losses = []
        for i in range(T):
            border_state = None if inference else batch.outputs.select_tensor_dim("timestep", i)
            for k in range(num_inter_steps):
                # maskedautoencoder strategy (lightning.py:580-581): the draw is the reference's (torch CPU generator, one per
                # model call), the product x * mask happens inside build_x (and its adjoint inside build_x's backward)
                blocks = None
                if self.mask_ratio != 0 and batch.forcing.tensor.dim() == 5:
                    blocks = ops.BlockMask.draw(batch.forcing.tensor.shape[2], batch.forcing.tensor.shape[3], self.mask_ratio, device)
                x = self._next_x(batch, prev_states, i, blocks=blocks, **x_format)
                if self.channels_last:
                    x = x.to(memory_format=torch.channels_last)
                if self.mask_ratio != 0 and blocks is None:
                    x = self.mask_tensor(x)   # graph layout: the reference's unpacking raises ValueError (3-d x), so does this
                if self.model.features_second:  # lightning.py:591-596
                    y = features_second_to_last(self.model(features_last_to_second(x)))
                elif x_format and hasattr(self.model, "rollout_padded_output"):
                    # the state update reads the first F features of rows of any width / float dtype: a model whose last kernel
                    # writes rows wider than out_channels (64-wide MLP / GEMM outputs) may hand them over as they are
                    self.model.rollout_padded_output = True
                    try:
                        y = call_model(x)
                    finally:
                        self.model.rollout_padded_output = False
                else:
                    y = call_model(x)

                last_prev = None if ds else prev_states.select_tensor_dim("timestep", -1)
                if ds:  # lightning.py:611-621: update the coarse forcing's common features
                    coarse = batch.forcing.select_tensor_dim("timestep", i)[..., self.common_features_idx]
                    last_prev, keep = coarse, 1.0
                else:
                    keep = keep_prev
                do_force = (not inference) and force_border
                if fuse:
                    new_state, loss_i = ops.ar_step_loss(
                        last_prev, y, border_state, std, mean, border_flat, interior_flat, f_weights, wl.num_interior,
                        f_count, wl.kind, f_mode, keep, do_force, out=f_states.select(1, i))
                    f_losses.append(loss_i)
                else:
                    new_state = ops.ar_update(
                        last_prev, y, border_state if do_force else None, std, mean,
                        border_flat if do_force else None, interior_flat if do_force else None,
                        keep_prev=keep, nan_to_num=self.mask_on_nan,
                    )
                if i < T - 1 or k < num_inter_steps - 1:  # lightning.py:636-656
                    t_dim = batch.inputs.dim_index("timestep")
                    if prev_states.dim_size("timestep") == 1:
                        new_prev = new_state.unsqueeze(t_dim)
                    else:
                        new_prev = torch.cat(
                            [prev_states.tensor.narrow(t_dim, 1, prev_states.dim_size("timestep") - 1),
                             new_state.unsqueeze(t_dim)], dim=t_dim)
                    prev_states = NamedTensor.new_like(new_prev, prev_states)
            prediction_list.append(new_state)

        if fuse:
            pred_out = NamedTensor.new_like(f_states.type_as(batch.outputs.tensor), batch.outputs)
            pred_out.fused_loss = torch.stack(f_losses, dim=1) * wl_weight
            if proxies is not None:
                proxies.attach(pred_out.fused_loss)
            return pred_out, batch.outputs
        prediction = torch.stack(prediction_list, dim=1)
        if proxies is not None:
            proxies.attach(prediction)
        if inference:
            pred_out = NamedTensor(prediction.type(self.output_dtype), self.output_dim_names, self.output_feature_names)
        else:
            pred_out = NamedTensor.new_like(prediction.type_as(batch.outputs.tensor), batch.outputs)
        return pred_out, batch.outputs

    def mask_tensor(self, x):
        """lightning.py:769-785 (MAE-style block masking) for callers outside the rollout; the rollout itself applies the same
        mask inside ``build_x``.  The reference clears, for every drawn index i, the block [row*bh, (row+1)*bh) x
        [col*bw, (col+1)*bw) with row = i // W, col = i % W, in a Python loop over up to H*W indices.  Equivalently: pixel
        (y, x) is cleared iff index (y // bh) * W + (x // bw) was drawn (same torch CPU generator draw, so the same mask bit
        for bit: tests/test_abi_cpu.py::test_mask_tensor_matches_reference_loop, tests/test_round2_gpu.py)."""
        _, height, width, _ = x.shape
        blocks = ops.BlockMask.draw(height, width, self.mask_ratio, x.device)
        return x * blocks.dense(height, width)[None, :, :, None]

    def get_mask_on_nan(self, target: NamedTensor):
        """
        lightning.py:787-797.  Returns (mask, target_masked).  ``mask`` is a lazy stand-in for the reference's tensor: the loss
        kernels of py4cast_amd.losses read it as a marker (they derive the mask from the target's NaNs, nothing is allocated);
        any other consumer -- plotters, metrics, ``mask * x``, ``mask.shape`` -- gets the reference's literal tensor, built on
        first use.  ``target_masked`` is likewise NaN-free only once somebody other than the loss kernels reads it.
        """
        if self.mask_on_nan:
            return NanMask(target.tensor), _LazyMaskedTarget(target)
        return OnesMask(target.tensor), target

    def materialize_mask(self, target: NamedTensor):
        """The reference's literal (mask tensor, NaN-free target) pair."""
        if self.mask_on_nan:
            mask = ~torch.isnan(target.tensor)
            t = target.clone()
            t.tensor = torch.nan_to_num(t.tensor, nan=0)
            return mask, t
        return torch.ones_like(target.tensor), target

    # ------------------------------------------------------------------ fit / val / test / predict
    def on_train_start(self):
        self.train_plotters = []

    def training_step(self, batch: ItemBatch, batch_idx: int) -> torch.Tensor:
        """lightning.py:806-831."""
        prediction, target = self.common_step(batch, batch_idx, phase="train")
        fused = getattr(prediction, "fused_loss", None)
        if fused is not None:  # native rollout already produced the (B,T) loss in its epilogue
            batch_loss = torch.mean(fused)
        else:
            mask, target_masked = self.get_mask_on_nan(target)
            batch_loss = torch.mean(self.loss(prediction, target_masked, mask=mask))
        self.training_step_losses.append(batch_loss.detach())
        return batch_loss

    def on_train_epoch_end(self):
        """lightning.py:833-839."""
        if self.logging_enabled and self.training_step_losses:
            experiment = getattr(getattr(self, "logger", None), "experiment", None)
            if experiment is not None and hasattr(experiment, "add_scalar"):
                experiment.add_scalar("mean_loss_epoch/train", torch.stack(list(self.training_step_losses)).mean(),
                                      getattr(self, "global_step", 0))
        self.training_step_losses.clear()

    # ------------------------------------------------------------------ housekeeping hooks of the reference (host side)
    def configure_loggers(self):
        """lightning.py:328-337."""
        layout = {"Check Overfit": {"loss": ["Multiline", ["mean_loss_epoch/train", "mean_loss_epoch/validation"]]}}
        experiment = getattr(getattr(self, "logger", None), "experiment", None)
        if experiment is not None and hasattr(experiment, "add_custom_scalars"):
            experiment.add_custom_scalars(layout)

    def print_summary_model(self):
        """lightning.py:399-414 (rank zero)."""
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_rank() != 0:
            return
        if hasattr(self.dataset_info, "summary"):
            self.dataset_info.summary()
        print(f"Number of input_steps : {self.num_input_steps}")
        print(f"Number of pred_steps (training) : {self.num_pred_steps_train}")
        print(f"Number of pred_steps (test/val) : {self.num_pred_steps_val_test}")
        print(f"Number of intermediary steps :{self.num_inter_steps}")
        print(f"Training strategy :{self.training_strategy}")
        print("---------------------")
        print(f"Loss {self.loss}")
        print(f"Batch size {self.batch_size}")
        print("---------------------------")
        n = sum(p.numel() for p in self.model.parameters())
        print(f"{type(self.model).__name__}: {n} parameters")

    def inspect_tensors(self):
        """lightning.py:416-428."""
        for name, param in self.named_parameters():
            print(name, param.shape, param.dtype)
        for name, buffer in self.named_buffers():
            print(name, buffer.shape, buffer.dtype)

    def log_hparams_tb(self):
        """lightning.py:430-448: the git state next to the logs (only when logging is on)."""
        if not self.logging_enabled:
            return
        import subprocess

        try:
            text = subprocess.check_output(["git", "log", "-n", "1"]).strip().decode() + \
                subprocess.check_output(["git", "status"]).strip().decode()
            (Path(self.trainer.logger.log_dir) / "git_log.txt").write_text(text)
        except Exception:   # not a git checkout: nothing to record
            pass

    def on_fit_start(self):
        """lightning.py:450-452."""
        self.log_hparams_tb()
        self.print_summary_model()

    def on_train_end(self):
        """lightning.py:841-858 logs the model to MLflow when an MLFlowLogger is attached; without one (this stack has no
        mlflow) there is nothing to do."""

    def load_weigths(self, path, map_location):
        """lightning.py:1109-1120 (sic): state dict with the "model." prefix removed from every key."""
        from collections import OrderedDict

        weights = torch.load(path, map_location)
        return OrderedDict((k.replace("model.", ""), v) for k, v in weights.items())

    # ------------------------------------------------------------------ observers (plots / metrics of the reference)
    # The plotters and the PSD metrics are py4cast's own host-side code (matplotlib, out of the hot-path scope): when the
    # `py4cast` package is importable they are attached and notified exactly where the reference notifies them
    # (lightning.py:868-1065); when it is not, the hooks are no-ops.  They receive the lazy mask / target of get_mask_on_nan.
    @staticmethod
    def _reference_observers():
        try:
            import py4cast.metrics as ref_metrics   # type: ignore
            import py4cast.plots as ref_plots       # type: ignore

            return ref_plots, ref_metrics
        except Exception:
            return None, None

    @property
    def current_epoch_(self) -> int:
        return int(getattr(self, "current_epoch", 0) or 0)

    def setup(self, stage=None):
        """lightning.py:312-326."""
        self.list_metrics = []
        if not self.logging_enabled:
            return
        self.save_path = Path(self.trainer.logger.log_dir)
        from .metrics import MetricACC   # device-side sums (p4c_acc_sums)

        self.acc_metric = MetricACC(self.dataset_info)
        self.list_metrics = [self.acc_metric]
        _, ref_metrics = self._reference_observers()
        if ref_metrics is not None:
            max_pred_step = self.num_pred_steps_val_test - 1
            self.rmse_psd_plot_metric = ref_metrics.MetricPSDVar(pred_step=max_pred_step)
            self.psd_plot_metric = ref_metrics.MetricPSDK(self.save_path, pred_step=max_pred_step)
            self.list_metrics += [self.psd_plot_metric, self.rmse_psd_plot_metric]

    def on_validation_start(self):
        """lightning.py:864-886."""
        self.valid_plotters = []
        ref_plots, _ = self._reference_observers()
        if self.logging_enabled and ref_plots is not None:
            l1_loss = ScaledLoss("L1Loss", reduction="none")
            l1_loss.prepare(self, self.interior_mask, self.dataset_info)
            sp = getattr(self, "save_path", None)
            self.valid_plotters = [
                ref_plots.StateErrorPlot({"mae": l1_loss}, prefix="Validation"),
                ref_plots.PredictionTimestepPlot(num_samples_to_plot=1, num_features_to_plot=4, prefix="Validation", save_path=sp),
                ref_plots.PredictionEpochPlot(num_samples_to_plot=1, num_features_to_plot=4, prefix="Validation", save_path=sp),
            ]

    def on_test_start(self):
        """lightning.py:986-1008."""
        self.test_plotters = []
        ref_plots, _ = self._reference_observers()
        if self.logging_enabled and ref_plots is not None:
            metrics = {}
            for torch_loss, alias in ("L1Loss", "mae"), ("MSELoss", "rmse"):
                loss = ScaledLoss(torch_loss, reduction="none")
                loss.prepare(self, self.interior_mask, self.dataset_info)
                metrics[alias] = loss
            sp = getattr(self, "save_path", None)
            self.test_plotters = [
                ref_plots.StateErrorPlot(metrics, save_path=sp),
                ref_plots.SpatialErrorPlot(),
                ref_plots.PredictionTimestepPlot(num_samples_to_plot=self.num_samples_to_plot, num_features_to_plot=4,
                                                 prefix="Test", save_path=sp),
            ]

    def _notify(self, plotters, batch, prediction, target, mask):
        for plotter in plotters:
            plotter.update(self, batch=batch, prediction=prediction, target=target, mask=mask)
        for metric in getattr(self, "list_metrics", []):
            if metric is getattr(self, "acc_metric", None):
                metric.update(prediction, target, mask)
            else:
                metric.update(prediction, target, mask, self.original_shape)

    def validation_step_logging(self, batch, prediction, target, mask):
        """lightning.py:919-941."""
        if self.logging_enabled:
            plot_period = 10   # PLOT_PERIOD, lightning.py:54
            self._notify(getattr(self, "valid_plotters", []) if self.current_epoch_ % plot_period == 0 else [], batch, prediction,
                         target, mask)

    def test_step_logging(self, batch, prediction, target, mask):
        """lightning.py:1044-1065."""
        if self.logging_enabled:
            self._notify(getattr(self, "test_plotters", []), batch, prediction, target, mask)

    def _epoch_end(self, plotters, label: str, prefix=None):
        """lightning.py:943-982 / 1067-1103: metric results are logged (tensors) or handed to the logger (figures)."""
        if not self.logging_enabled:
            return
        results = {}
        for metric in getattr(self, "list_metrics", []):
            results.update(metric.compute() if prefix is None else metric.compute(prefix=prefix))
        figures = {k: v for k, v in results.items() if not isinstance(v, torch.Tensor)}
        self.log_dict({k: v for k, v in results.items() if isinstance(v, torch.Tensor)}, prog_bar=False, on_step=False,
                      on_epoch=True, sync_dist=True)
        experiment = getattr(getattr(self, "logger", None), "experiment", None)
        if experiment is not None and hasattr(experiment, "add_figure"):
            for name, fig in figures.items():
                experiment.add_figure(f"{name}", fig, self.current_epoch_)
        for plotter in plotters:
            plotter.on_step_end(self, label=label)

    def on_validation_epoch_end(self):
        self._epoch_end(getattr(self, "valid_plotters", []) if self.current_epoch_ % 10 == 0 else [], "Valid")
        self.validation_step_losses.clear()

    def on_test_epoch_end(self):
        self._epoch_end(getattr(self, "test_plotters", []), "Test", prefix="test")

    def _eval_step(self, batch: ItemBatch, batch_idx: int, label: str):
        with torch.no_grad():
            prediction, target = self.common_step(batch, batch_idx, phase="val_test")
            mask, target_masked = self.get_mask_on_nan(target)
            fused = getattr(prediction, "fused_loss", None)   # (B,T) loss already produced by the rollout's fused steps
            loss_bt = fused if fused is not None else self.loss(prediction, target_masked, mask)
            time_step_loss = torch.mean(loss_bt, dim=0)  # lightning.py:895
            mean_loss = torch.mean(time_step_loss)
        return prediction, target_masked, mask, time_step_loss, mean_loss

    def validation_step(self, batch: ItemBatch, batch_idx: int):
        """lightning.py:888-917."""
        prediction, target_masked, mask, time_step_loss, mean_loss = self._eval_step(batch, batch_idx, "val")
        log = {"val_mean_loss": mean_loss}
        for step in range(time_step_loss.shape[0]):
            log[f"val_loss_step_{step + 1}"] = time_step_loss[step]
        self.log_dict(log, sync_dist=True)
        self.validation_step_losses.append(mean_loss)
        self.val_mean_loss = mean_loss
        self.validation_step_logging(batch, prediction, target_masked, mask)
        return mean_loss

    def test_step(self, batch: ItemBatch, batch_idx: int):
        """lightning.py:1017-1042."""
        prediction, target_masked, mask, time_step_loss, mean_loss = self._eval_step(batch, batch_idx, "test")
        log = {"test_mean_loss": mean_loss}
        for step in range(time_step_loss.shape[0]):
            log[f"test_loss_step_{step + 1}"] = time_step_loss[step]
        self.log_dict(log, sync_dist=True)
        self.test_step_logging(batch, prediction, target_masked, mask)
        return mean_loss

    def predict_step(self, batch: ItemBatch, batch_idx: int) -> torch.Tensor:
        """lightning.py:1118-1188: rollout without border forcing, then un-normalise per feature (:1162-1169).  With an
        ``output_stager`` attached (py4cast_amd.outputs.OutputStager) the un-normalised prediction also leaves for the host as
        feature-major planes in a pinned buffer -- what the GRIB / GIF writers of io/outputs.py consume -- without blocking;
        ``self.staged_slot`` is the ticket for ``output_stager.wait``."""
        if batch_idx == 0 and getattr(self, "input_feature_names", None) is not None:
            if self.input_feature_names != batch.inputs.feature_names:   # lightning.py:1123-1128
                raise ValueError(
                    f"Input Feature names mismatch between training and inference. "
                    f"Training: {self.input_feature_names}, Inference: {batch.inputs.feature_names}"
                )
        with torch.no_grad():
            preds = self.forward(batch, batch_idx)
            std = self.stats.to_list("std", preds.feature_names).to(preds.tensor)
            mean = self.stats.to_list("mean", preds.feature_names).to(preds.tensor)
            stager = getattr(self, "output_stager", None)
            if stager is not None:
                self.staged_slot = stager.submit(preds, std, mean)   # reads the normalised tensor: before the in-place pass below
            t = preds.tensor.contiguous().float()
            preds.tensor = ops.unnormalize(t, std, mean, out=t)  # one kernel, the reference's two rounded steps
        return preds
