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
        if self._views_ok and self._grads_are_views():
            self.flat_grad.zero_()
        else:
            for p in self.params:
                if p.grad is not None:
                    p.grad.zero_()

    def _grads_are_views(self) -> bool:
        """Every ``p.grad`` still IS its slice of the flat bucket (same address, so same storage at the expected offset): a grad
        re-created by autograd after ``zero_grad(set_to_none=True)`` -- wherever the allocator put it -- fails this."""
        base, off = self.flat_grad.data_ptr(), 0
        for p in self.params:
            g = p.grad
            if g is None or g.dtype != torch.float32 or g.data_ptr() != base + 4 * off or not g.is_contiguous():
                return False
            off += p.numel()
        return True

    def all_reduce_grads(self):
        if self.world_size <= 1:
            return
        if not self._views_ok or not self._grads_are_views():
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


class GraphedTrainingStep:
    """One micro-batch -- ``training_step`` (rollout + loss) and its ``backward`` -- captured in a HIP graph and replayed.

    The eager step of a small-kernel model is bound by the host, not the GPU: a hierarchical GNN issues ~10^4 launches per step
    from Python (HiLAM at 512x512: 164 ms eager per step for ~30 ms of kernels).  Every entry point of the C ABI only enqueues on
    the caller's stream and takes its work-spaces from the caller, so a whole step is capturable; replaying it costs one launch.
    Inputs live in static tensors (a replay copies the new batch in), gradients accumulate into the parameters' existing ``.grad``
    buffers (FlatDDP's flat bucket), so ``all_reduce_grads`` / ``optimizer.step`` / ``zero_grad`` run outside the graph as usual.
    Shapes must not change between steps; a module whose step synchronises with the host cannot be captured (the constructor
    raises, nothing is left half-captured)."""

    def __init__(self, module, sample_batch, loss_scale: float = 1.0, warmup: int = 3):
        from .base import ItemBatch
        from .namedtensor import NamedTensor

        if not torch.cuda.is_available():
            raise RuntimeError("GraphedTrainingStep needs a GPU")
        self.module = module
        self._static = {}
        for name in ("inputs", "forcing", "outputs"):
            nt = getattr(sample_batch, name)
            self._static[name] = (nt.tensor.clone(), list(nt.names), list(nt.feature_names))

        def fresh():   # graph models flatten the batch's NamedTensors in place (lightning.py:526-535): new wrappers every call
            return ItemBatch(*[NamedTensor(t, list(n), list(f)) for t, n, f in (self._static[k] for k in ("inputs", "forcing", "outputs"))])

        def run(idx):
            loss = module.training_step(fresh(), idx)
            (loss * loss_scale if loss_scale != 1.0 else loss).backward()
            return loss.detach()

        # NOTE for callers: no tensor of an earlier EAGER step of this module may still be referenced here (typically its loss):
        # it keeps that step's AccumulateGrad nodes alive, they are bound to the stream they were created on (the default stream),
        # and autograd would make that stream wait on the capturing one -- which breaks the capture (observed: a crash in the HIP
        # runtime).  Keep `loss.detach()` / `float(loss)` instead, as `AutoRegressiveLightning.training_step_losses` does.
        import gc

        gc.collect()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(warmup):      # lazy initialisation (edge sets, kernel attributes, allocator pools) happens here
                run(i)
        torch.cuda.current_stream().wait_stream(side)
        from . import _lib as L

        self.graph = torch.cuda.CUDAGraph()
        L.CAPTURE_SCOPE[0] = {}
        try:
            with torch.cuda.graph(self.graph):
                self.loss = run(warmup)
        finally:
            L.CAPTURE_SCOPE[0] = None
        self.warmup_backwards = warmup + 1   # gradient contributions already accumulated by construction: zero_grad() after

    def accepts(self, batch) -> bool:
        """The captured graph serves batches of the captured shapes only."""
        return all(getattr(batch, n).tensor.numel() == self._static[n][0].numel()
                   and getattr(batch, n).tensor.shape[0] == self._static[n][0].shape[0] for n in ("inputs", "forcing", "outputs"))

    def __call__(self, batch):
        for name in ("inputs", "forcing", "outputs"):
            src = getattr(batch, name).tensor
            dst = self._static[name][0]
            dst.copy_(src.reshape(dst.shape), non_blocking=True)
        self.graph.replay()
        return self.loss


class _Logger:
    log_dir = None


class Trainer:
    """Subset of lightning.Trainer used by AutoRegressiveLightning (config/CLI/trainer.yaml)."""

    def __init__(self, max_epochs: int = 1, max_steps: int = -1, accumulate_grad_batches: int = 1,
                 precision: str = "32-true", limit_train_batches: Optional[int] = None,
                 limit_val_batches: Optional[int] = None, device: Optional[torch.device] = None,
                 fast_dev_run: bool = False, hip_graph: Optional[bool] = None):
        self.max_epochs, self.max_steps = (1, 1) if fast_dev_run else (max_epochs, max_steps)
        self.accumulate_grad_batches = accumulate_grad_batches
        self.precision = {"32": "32-true", 32: "32-true", "bf16": "bf16-true"}.get(precision, precision)
        self.limit_train_batches, self.limit_val_batches = limit_train_batches, limit_val_batches
        self.device = device or (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        self.logger = _Logger()
        self.hip_graph = hip_graph   # None: follow the model's `prefers_hip_graph`; True / False: force
        self.global_step = 0
        self.estimated_stepping_batches = 1000
        self.callback_metrics = {}
        self.train_step_losses = []   # detached per-micro-batch training losses of the last fit (device tensors: no sync here)

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
        self.train_step_losses = []
        if hasattr(module, "on_train_start"):
            module.on_train_start()
        ddp.zero_grad()
        done = False
        use_graph = self.device.type == "cuda" and (self.hip_graph if self.hip_graph is not None
                                                    else getattr(getattr(module, "model", None), "prefers_hip_graph", False))
        graphed = None
        for epoch in range(self.max_epochs):
            module.train()
            pending = 0   # micro-batches accumulated since the last optimizer step
            it = iter(train_dataloader)
            nxt = next(it, None)
            i = -1
            while nxt is not None:
                batch, nxt = nxt, next(it, None)
                i += 1
                if self.limit_train_batches and i >= self.limit_train_batches:
                    break
                last_of_epoch = nxt is None or bool(self.limit_train_batches and i + 1 >= self.limit_train_batches)
                batch = self._to_device(batch, self.device)
                loss = None   # see GraphedTrainingStep: nothing of an earlier eager step may be alive at capture time
                if graphed is None and use_graph and i > 0:
                    # the first micro-batch ran eagerly (names / dtypes recorded, lazy initialisation done); capture from the second.
                    # The capture's warm-up passes are not part of the training run: gradients and module buffers (BatchNorm
                    # running statistics, num_batches_tracked) are restored afterwards
                    saved = ddp.flat_grad.clone()
                    buffers = [(b, b.detach().clone()) for b in module.buffers()]
                    graphed = GraphedTrainingStep(module, batch, loss_scale=1.0 / self.accumulate_grad_batches)
                    ddp.flat_grad.copy_(saved)
                    for b, keep in buffers:
                        b.copy_(keep)
                if graphed is not None and graphed.accepts(batch):
                    loss = graphed(batch)
                else:   # eager; also a batch whose shape differs from the captured one (a short last batch)
                    loss = module.training_step(batch, i)
                    (loss / self.accumulate_grad_batches).backward()
                self.train_step_losses.append(loss.detach())
                pending += 1
                # Lightning steps every accumulate_grad_batches micro-batches AND on the last batch of an epoch: leftovers do not
                # leak into the next epoch.  Non-stepping micro-batches do not sync
                if pending == self.accumulate_grad_batches or last_of_epoch:
                    pending = 0
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
