"""
Minimal fit loop + data-parallel gradient exchange for the hot path.

The reference trains through ``lightning.Trainer`` (config/CLI/trainer.yaml): DDP strategy = one
process per GPU, one gradient all-reduce per optimizer step, ``accumulate_grad_batches`` micro
batches per optimizer step with no sync on the non-stepping ones (trainer.yaml:58).  When
``lightning`` is installed, ``py4cast_amd.lightning.AutoRegressiveLightning`` is driven by it
unchanged.  This module provides the same loop without Lightning (the build image has none):

* ``FlatDDP``: all parameter gradients live in ONE flat fp32 buffer (each ``param.grad`` is a view
  into it), so the exchange is a single RCCL all-reduce over xGMI per optimizer step.  With BPTT
  every gradient is final only once the backward of AR step 0 has finished, and HalfUNet-sized
  models have ~2 MB of gradients: one latency-bound collective, launched right after backward.
* ``Trainer``: hook-compatible subset (``training_step``, ``validation_step``,
  ``configure_optimizers``, ``on_train_start`` ...), gradient accumulation, per-step LR schedule.
"""

from typing import Iterable, Optional

import torch
import torch.distributed as dist


class FlatDDP:
    """Flat-bucket gradient all-reduce (mean over ranks) for one module."""

    def __init__(self, module: torch.nn.Module, world_size: Optional[int] = None):
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.world_size = world_size
        self.params = [p for p in module.parameters() if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.dtype == torch.float32:
                p.grad = self.flat_grad[off : off + n].view_as(p)
            off += n
        self._views_ok = all(p.dtype == torch.float32 for p in self.params)
        if world_size > 1:
            self.broadcast_parameters()

    def broadcast_parameters(self, src: int = 0):
        """Same initial weights on every rank (DDP's constructor does the same)."""
        for p in self.params:
            dist.broadcast(p.data, src)

    def zero_grad(self):
        """One fill over the flat buffer instead of one per parameter (``optimizer.zero_grad(set_to_none=False)``)."""
        base = self.flat_grad.untyped_storage().data_ptr()
        if self._views_ok and all(p.grad is not None and p.grad.untyped_storage().data_ptr() == base for p in self.params):
            self.flat_grad.zero_()
        else:
            for p in self.params:
                if p.grad is not None:
                    p.grad.zero_()

    def all_reduce_grads(self):
        if self.world_size <= 1:
            return
        if not self._views_ok or any(p.grad is None or p.grad.data_ptr() < self.flat_grad.data_ptr() for p in self.params):
            self._regather()
        dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM)
        self.flat_grad.div_(self.world_size)
        if not self._views_ok:
            self._scatter()

    # slow path: a grad tensor was replaced (e.g. zero_grad(set_to_none=True)); copy in / re-attach
    def _regather(self):
        off = 0
        for p in self.params:
            n = p.numel()
            view = self.flat_grad[off : off + n].view_as(p)
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
            if p.dtype == torch.float32:
                p.grad = view
            off += n

    def _scatter(self):
        off = 0
        for p in self.params:
            n = p.numel()
            if p.dtype != torch.float32:
                p.grad.copy_(self.flat_grad[off : off + n].view_as(p))
            off += n


class _Logger:
    log_dir = None


class Trainer:
    """Subset of lightning.Trainer used by AutoRegressiveLightning (config/CLI/trainer.yaml)."""

    def __init__(self, max_epochs: int = 1, max_steps: int = -1, accumulate_grad_batches: int = 1,
                 precision: str = "32-true", limit_train_batches: Optional[int] = None,
                 limit_val_batches: Optional[int] = None, device: Optional[torch.device] = None,
                 fast_dev_run: bool = False):
        self.max_epochs, self.max_steps = (1, 1) if fast_dev_run else (max_epochs, max_steps)
        self.accumulate_grad_batches = accumulate_grad_batches
        self.precision = {"32": "32-true", 32: "32-true", "bf16": "bf16-true"}.get(precision, precision)
        self.limit_train_batches, self.limit_val_batches = limit_train_batches, limit_val_batches
        self.device = device or (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        self.logger = _Logger()
        self.global_step = 0
        self.estimated_stepping_batches = 1000
        self.callback_metrics = {}

    @staticmethod
    def _to_device(batch, device):
        for name in ("inputs", "forcing", "outputs"):
            nt = getattr(batch, name, None)
            if nt is not None:
                nt.tensor = nt.tensor.to(device, non_blocking=True)
        return batch

    def fit(self, module, train_dataloader: Iterable, val_dataloader: Optional[Iterable] = None):
        module.trainer = self
        module.to(self.device)
        try:
            n_batches = len(train_dataloader)
            if self.limit_train_batches:
                n_batches = min(n_batches, self.limit_train_batches)
            self.estimated_stepping_batches = max(1, (n_batches * self.max_epochs) // self.accumulate_grad_batches)
        except TypeError:
            pass
        if self.max_steps > 0:
            self.estimated_stepping_batches = self.max_steps
        conf = module.configure_optimizers()
        opt, sched = conf["optimizer"], conf.get("lr_scheduler", {}).get("scheduler")
        ddp = FlatDDP(module)
        if hasattr(module, "on_train_start"):
            module.on_train_start()
        ddp.zero_grad()
        done = False
        for epoch in range(self.max_epochs):
            module.train()
            for i, batch in enumerate(train_dataloader):
                if self.limit_train_batches and i >= self.limit_train_batches:
                    break
                loss = module.training_step(self._to_device(batch, self.device), i)
                (loss / self.accumulate_grad_batches).backward()
                if (i + 1) % self.accumulate_grad_batches == 0:  # non-stepping micro-batches do not sync
                    ddp.all_reduce_grads()
                    opt.step()
                    if sched is not None:
                        sched.step()
                    ddp.zero_grad()
                    self.global_step += 1
                    if 0 < self.max_steps <= self.global_step:
                        done = True
                        break
            if hasattr(module, "on_train_epoch_end"):
                module.on_train_epoch_end()
            if val_dataloader is not None:
                self.validate(module, val_dataloader)
            if done:
                break
        return module

    def validate(self, module, dataloader: Iterable):
        module.trainer = self
        module.eval()
        losses = []
        for i, batch in enumerate(dataloader):
            if self.limit_val_batches and i >= self.limit_val_batches:
                break
            losses.append(module.validation_step(self._to_device(batch, self.device), i))
        if losses:
            mean = torch.stack([torch.as_tensor(l) for l in losses]).mean()
            if dist.is_available() and dist.is_initialized():  # self.log(..., sync_dist=True) (lightning.py:904-911)
                dist.all_reduce(mean, op=dist.ReduceOp.SUM)
                mean /= dist.get_world_size()
            self.callback_metrics["val_mean_loss"] = mean
        return self.callback_metrics
