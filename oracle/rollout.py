"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement, in plain torch ops on raw tensors, of the autoregressive rollout of
``/root/reference/py4cast/lightning.py``.  No NamedTensor, no Lightning.

Tensor conventions (grid layout): inputs (B,T_in,H,W,F), forcing (B,T,H,W,Ff),
outputs (B,T,H,W,F), statics (B,H,W,Fs), border/interior masks (H,W,1).
Graph layout: the two spatial dims are flattened into one (lightning.py:526-535);
every function below is written on "...spatial..., features" so both work.
"""

from typing import Callable, List, Optional, Sequence, Tuple

import torch


def common_features_idx(output_feature_names: Sequence[str], forcing_feature_names: Sequence[str]) -> List[int]:
    """lightning.py:548-557 (downscaling_only): forcing indices whose name matches an output name after the first
    underscore-separated token."""
    idx = []
    for out_name in output_feature_names:
        for i, forcing_name in enumerate(forcing_feature_names):
            if out_name.split("_")[1:] == forcing_name.split("_")[1:]:
                idx.append(i)
    return idx


def cosine_with_min_lr(step: int, num_warmup_steps: int, num_training_steps: int, min_lr_rate: float,
                       num_cycles: float = 0.5) -> float:
    """The LR multiplier of transformers' get_cosine_with_min_lr_schedule_with_warmup (third-party, pinned
    transformers==4.45.2 in the reference's requirements; used at lightning.py:453-458 with min_lr=min_learning_rate, i.e.
    min_lr_rate = min_lr / lr).  Published algorithm: linear warm-up, then a cosine from 1 to min_lr_rate."""
    import math

    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    factor = 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress))
    factor = factor * (1 - min_lr_rate) + min_lr_rate
    return max(0, factor)


def strategy_params(training_strategy: str, num_inter_steps: int) -> Tuple[bool, bool, int]:
    """lightning.py:678-694 (_strategy_params)."""
    force_border = training_strategy == "scaled_ar"
    scale_y = training_strategy == "scaled_ar"
    if training_strategy == "diff_ar" and num_inter_steps != 1:
        raise ValueError("Diff AR strategy requires exactly 1 intermediary step.")
    return force_border, scale_y, num_inter_steps


def next_x(
    prev_states: torch.Tensor,
    statics: torch.Tensor,
    forcing_i: torch.Tensor,
    num_input_steps: int,
    mask_on_nan: bool = False,
    downscaling_only: bool = False,
) -> torch.Tensor:
    """
    lightning.py:711-767 (_next_x).  Channel order: prev_states[:,0..T_in-1], statics,
    forcing(step i), then the optional "not NaN anywhere" mask channel.
    ``forcing_i`` is the already time-selected forcing (B,...,Ff).
    """
    inputs = [prev_states.select(1, t) for t in range(num_input_steps)]
    mask_list = []
    if mask_on_nan:
        # lightning.py:732-757: union over every input & forcing channel (bool ops, bit exact)
        combined = torch.zeros_like(inputs[0][..., 0], dtype=torch.bool)
        for inp in inputs:
            combined = combined | torch.isnan(inp).any(dim=-1)
        combined = combined | torch.isnan(forcing_i).any(dim=-1)
        mask_list.append(~combined.unsqueeze(-1))
        inputs = [torch.nan_to_num(inp, nan=0) for inp in inputs]
        forcing_i = torch.nan_to_num(forcing_i, nan=0)
    parts = ([] if downscaling_only else inputs) + [statics[: prev_states.shape[0]], forcing_i] + mask_list
    return torch.cat(parts, dim=-1)


def rollout(
    model_fn: Callable[[torch.Tensor], torch.Tensor],
    inputs: torch.Tensor,
    forcing: torch.Tensor,
    outputs: Optional[torch.Tensor],
    statics: torch.Tensor,
    border_mask: torch.Tensor,
    interior_mask: torch.Tensor,
    diff_std: Optional[torch.Tensor],
    diff_mean: Optional[torch.Tensor],
    training_strategy: str = "scaled_ar",
    num_inter_steps: int = 1,
    mask_on_nan: bool = False,
    phase: str = "train",
    features_second: bool = False,
    num_pred_steps: Optional[int] = None,
    common_features_idx: Optional[Sequence[int]] = None,
    mask_ratio: float = 0,
) -> torch.Tensor:
    """
    lightning.py:495-676 (_common_step).  ``model_fn`` maps x -> y in the layout the model
    declares (``features_second`` => (B,C,H,W), lightning.py:591-596).  Returns the stacked
    prediction (B,T,...,F).
    """
    force_border, scale_y, K = strategy_params(training_strategy, num_inter_steps)
    ds = training_strategy == "downscaling_only"
    inference = phase == "inference"
    T = num_pred_steps if num_pred_steps is not None else outputs.shape[1]
    T_in = inputs.shape[1]
    prev_states = inputs
    preds: List[torch.Tensor] = []
    for i in range(T):
        if not inference:
            border_state = outputs.select(1, i).clone()  # :567
            if mask_on_nan:
                border_state = torch.nan_to_num(border_state, nan=0)
        for k in range(K):
            x = next_x(prev_states, statics, forcing.select(1, i), T_in, mask_on_nan, ds)
            if mask_ratio != 0:  # :580-581, one draw from the global CPU generator per model call (:775)
                n_blocks = int((1 - mask_ratio) * x.shape[1] * x.shape[2])
                x = mask_tensor(x, mask_ratio, torch.randperm(x.shape[1] * x.shape[2])[:n_blocks])
            if features_second:
                y = model_fn(x.movedim(-1, 1)).movedim(1, -1)
            else:
                y = model_fn(x)
            last_prev = prev_states.select(1, -1).clone()  # :599
            if mask_on_nan:
                last_prev = torch.nan_to_num(last_prev, nan=0)
            if scale_y:  # :604-610
                predicted = last_prev * (1 - ds) + y * diff_std + diff_mean
            elif ds:  # :611-621
                coarse = forcing.select(1, i).clone()
                if mask_on_nan:
                    coarse = torch.nan_to_num(coarse, nan=0)
                predicted = coarse[..., list(common_features_idx)] + y
            else:  # :623
                predicted = last_prev * (1 - ds) + y
            if (not inference) and force_border:  # :627-633
                new_state = border_mask * border_state + interior_mask * predicted
            else:
                new_state = predicted
            if i < T - 1 or k < K - 1:  # :636-656
                prev_states = torch.cat([prev_states[:, 1:], new_state.unsqueeze(1)], dim=1)
        preds.append(new_state)  # :658 (outside the k loop)
    prediction = torch.stack(preds, dim=1)  # :660
    if outputs is not None:
        prediction = prediction.type_as(outputs)  # :674
    return prediction


def get_mask_on_nan(target: torch.Tensor, mask_on_nan: bool):
    """lightning.py:787-797."""
    if mask_on_nan:
        return ~torch.isnan(target), torch.nan_to_num(target, nan=0)
    return torch.ones_like(target), target


def mask_tensor(x: torch.Tensor, mask_ratio: float, block_indices: torch.Tensor) -> torch.Tensor:
    """
    lightning.py:769-785 with the random permutation factored out: ``block_indices`` is
    ``torch.randperm(H*W)[:int((1-mask_ratio)*H*W)]`` (the reference draws it from the
    global CPU generator).
    """
    _, height, width, _ = x.shape
    bh = height // int(height**0.5)
    bw = width // int(width**0.5)
    mask = torch.ones_like(x, dtype=torch.bool)
    for i in block_indices.tolist():
        row, col = i // width, i % width
        mask[:, row * bh : (row + 1) * bh, col * bw : (col + 1) * bw, :] = False
    return x * mask
