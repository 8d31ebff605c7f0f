"""
``FlatAdamW``: torch.optim.AdamW (the optimizer of ``configure_optimizers``, py4cast/lightning.py:442-467) whose step is ONE
HIP kernel when all parameters -- and all their gradients -- are views of one flat fp32 buffer each, in the same order
(the layout HalfUNetMI355X / FlatDDP set up).  It *is* a ``torch.optim.AdamW`` (same constructor, param groups, LR-scheduler
and ``state_dict`` behaviour: the per-parameter ``exp_avg`` / ``exp_avg_sq`` / ``step`` entries exist and alias the flat
state); anything it cannot take on the flat path (CPU tensors, scattered storages, amsgrad, maximize, several groups) goes
through the parent's step unchanged.
"""

from typing import List, Optional

import torch

from . import _lib as L


def _flat_view(tensors: List[torch.Tensor]) -> Optional[torch.Tensor]:
    """The flat fp32 tensor the given tensors tile, in order and without gaps, or None."""
    t0 = tensors[0]
    if t0.dtype != torch.float32 or not t0.is_cuda:
        return None
    ptr, total = t0.data_ptr(), 0
    for t in tensors:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() != ptr + 4 * total or t.device != t0.device:
            return None
        total += t.numel()
    store = t0.untyped_storage()
    first = (ptr - store.data_ptr()) // 4
    if ptr < store.data_ptr() or (first + total) * 4 > store.nbytes():
        return None
    return torch.empty(0, dtype=torch.float32, device=t0.device).set_(store, first, (total,))


class FlatAdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kw)
        self._flat_state = None  # (exp_avg, exp_avg_sq) flat buffers the per-parameter state entries alias

    def _flat_buffers(self, group):
        params = [p for p in group["params"] if p.requires_grad]
        if not params or any(p.grad is None or p.grad.is_sparse for p in params):
            return None
        flat_p = _flat_view([p.data for p in params])
        flat_g = _flat_view([p.grad for p in params]) if flat_p is not None else None
        if flat_p is None or flat_g is None:
            return None
        if self._flat_state is None or self._flat_state[0].numel() != flat_p.numel() or self._flat_state[0].device != flat_p.device:
            m, v = torch.zeros_like(flat_p), torch.zeros_like(flat_p)
            off = 0
            for p in params:   # carry over an existing (e.g. loaded) per-parameter state, then alias it to the flat state
                st = self.state[p]
                n = p.numel()
                if "exp_avg" in st:
                    m[off : off + n].copy_(st["exp_avg"].reshape(-1))
                    v[off : off + n].copy_(st["exp_avg_sq"].reshape(-1))
                st["exp_avg"], st["exp_avg_sq"] = m[off : off + n].view_as(p), v[off : off + n].view_as(p)
                if "step" not in st:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                off += n
            self._flat_state = (m, v)
        else:
            m, v = self._flat_state
            off = 0
            for p in params:   # somebody replaced a state tensor (load_state_dict): fall back for safety
                st = self.state[p]
                if "exp_avg" not in st or st["exp_avg"].data_ptr() != m.data_ptr() + 4 * off:
                    self._flat_state = None
                    return self._flat_buffers(group)
                off += p.numel()
        return params, flat_p, flat_g, m, v

    @torch.no_grad()
    def step_shards(self, shards):
        """Data-parallel sharded step (trainer.FlatDDP(sharded=True)): update only the flat ranges ``[(lo, hi), ...]`` this rank owns
        (their gradients are the reduce-scattered means); the caller all-gathers the parameters afterwards.  Needs the flat layout."""
        return self.step(shards=list(shards))

    @torch.no_grad()
    def step(self, closure=None, shards=None):
        if shards is not None and len(self.param_groups) != 1:
            raise RuntimeError("FlatAdamW.step_shards needs a single parameter group")
        if len(self.param_groups) != 1:
            return super().step(closure)
        group = self.param_groups[0]
        if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable"):
            return super().step(closure)
        bufs = self._flat_buffers(group)
        if bufs is None:
            if shards is not None:
                raise RuntimeError("FlatAdamW.step_shards: parameters and gradients must be views of flat buffers on the GPU")
            self._flat_state = None
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        params, flat_p, flat_g, m, v = bufs
        steps = {float(self.state[p]["step"]) for p in params}
        if len(steps) != 1:
            self._flat_state = None
            return super().step(closure)
        step = int(steps.pop()) + 1
        beta1, beta2 = group["betas"]
        for lo, hi in (shards if shards is not None else [(0, flat_p.numel())]):
            hi = min(hi, flat_p.numel())   # (FlatDDP pads its buffers to a multiple of the world size: the tail is nobody's)
            if hi > lo:
                L.call("p4c_adamw_step", L.ptr(flat_p[lo:hi]), L.ptr(flat_g[lo:hi]), L.ptr(m[lo:hi]), L.ptr(v[lo:hi]), hi - lo,
                       float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]), step,
                       L.stream(flat_p.device))
        for p in params:
            self.state[p]["step"] += 1
        L.PARAM_EPOCH[0] += 1   # the kernel wrote the parameters behind autograd's back (tensor._version did not move)
        return loss
