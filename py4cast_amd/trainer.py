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


class _GradSinkListener:
    """FlatDDP's entry in ``_lib.GRAD_SINK_LISTENERS`` (a weak reference: a dropped FlatDDP stops listening by itself)."""

    def __init__(self, owner):
        import weakref

        self.ref = weakref.ref(owner, lambda _r: self.remove())

    def remove(self):
        from . import _lib as L

        if self in L.GRAD_SINK_LISTENERS:
            L.GRAD_SINK_LISTENERS.remove(self)

    def sink_taken(self, view):
        owner = self.ref()
        if owner is not None:
            owner._sink_taken(view)

    def written(self, views):
        owner = self.ref()
        if owner is not None:
            owner._written(views)


class FlatDDP:
    """Flat-bucket gradient exchange (mean over ranks) for one module.

    * every ``param.grad`` is a view of ONE flat fp32 buffer, so an exchange moves whole buckets, never single tensors;
    * the collectives are issued on a COMMUNICATION stream that waits for the backward already enqueued on the compute stream;
      the compute stream is made to wait for them only when ``wait()`` / the next kernel that needs the gradients comes, so the
      host goes on enqueueing (the optimizer's launch, the next batch's copies) while xGMI is busy;
    * small models (HalfUNet: 1.8 MB) exchange ONE bucket -- latency-bound, a ring per xGMI link cannot help; above
      ``single_bucket_bytes`` the flat buffer is cut into ``bucket_bytes`` pieces issued back to back (several collectives in
      flight keep all seven links of a GPU busy), last layers first (the order their gradients become final in the backward of
      AR step 0 -- with BPTT nothing is final earlier);
    * ``sharded=True`` (large models, e.g. UNetR++): reduce-scatter instead of all-reduce, the optimizer steps only this rank's
      shard of every bucket (``shards()``; ``FlatAdamW.step(shards=...)``), then ``all_gather_params`` -- the same bytes on the
      links as an all-reduce, 1/N of the optimizer work per rank;
    * ``overlap=True`` (several buckets, parameters that go through autograd): the exchange starts INSIDE the backward
      (config/CLI/trainer.yaml:58,62-64 -- what Lightning's DDP does with its reducer).  With BPTT autograd sums a parameter's
      contributions of all AR steps before its AccumulateGrad node runs, i.e. a gradient becomes final while the backward of AR
      step 0 passes its layer (the last 1/T of the sweep): a post-accumulate hook counts the parameters of each bucket, and a
      bucket whose count is complete is issued on the communication stream at once -- buckets strictly in the order last ->
      first, whatever order the hooks fire in, so that every rank issues the same sequence of collectives.  ``arm()`` before
      the backward of the micro-batch that steps; ``all_reduce_grads()`` afterwards issues what is left and waits.
    * gradients written IN PLACE (``ops_gemm.GRADS_IN_PLACE`` / ``graphlam.GRADS_IN_PLACE``: a reduction kernel adds dW into the
      ``.grad`` view, autograd sees None -- UNETR++'s 100+ MB of them) pass no AccumulateGrad node and fire no hook.  The ops
      report them instead (``_lib.grad_sink_taken`` in the forward, ``_lib.grad_written`` right after the adding kernel is
      enqueued in the backward).  With BPTT such a parameter receives one write per use; how many is LEARNED: the first armed
      backward after a parameter's gradient was seen to go in place only counts (writes, hook firings) and keeps the parameter's
      buckets back; from the next one on the parameter is final at its last expected event and its bucket leaves inside the
      backward like the others (what ``DistributedDataParallel(static_graph=True)`` does with its first iteration).  A pass with
      FEWER events than learned leaves the bucket to ``all_reduce_grads`` and re-learns; an event that arrives for a bucket already
      on the wire (a program whose use counts grow from step to step) raises inside the backward -- never a silent wrong mean.
    """

    def __init__(self, module: torch.nn.Module, world_size: Optional[int] = None, bucket_bytes: int = 64 << 20,
                 single_bucket_bytes: int = 16 << 20, sharded: bool = False, overlap: bool = False):
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.world_size = world_size
        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.params = [p for p in module.parameters() if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        pad = (-total) % max(world_size, 1)    # shards of equal size: the flat buffers carry a few unused elements at the end
        self.total = total
        self.flat_grad = torch.zeros(total + pad, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.dtype == torch.float32:
                p.grad = self.flat_grad[off : off + n].view_as(p)
            off += n
        self._views_ok = all(p.dtype == torch.float32 for p in self.params)
        self.sharded = bool(sharded) and world_size > 1
        # buckets: [lo, hi) element ranges of the flat buffer, each a multiple of world_size long
        n_all = total + pad
        if n_all * 4 <= single_bucket_bytes or world_size <= 1:
            self.buckets = [(0, n_all)]
        else:
            per = max(world_size, (bucket_bytes // 4) // max(world_size, 1) * max(world_size, 1))
            self.buckets = [(lo, min(lo + per, n_all)) for lo in range(0, n_all, per)]
        self.comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._inflight = False
        # ---- exchange overlapped with the backward (see the class docstring)
        self.overlap = bool(overlap) and world_size > 1 and len(self.buckets) > 1 and self._views_ok
        self._armed = False
        self._issued = [False] * len(self.buckets)       # bucket already on the communication stream in this exchange
        self._pending = [0] * len(self.buckets)          # parameters of the bucket whose gradient is not final yet
        self.issued_in_backward = 0                      # (diagnostics / tests) buckets issued by hooks in the last armed backward
        self._bucket_params = [0] * len(self.buckets)
        self._hooks = []
        self._owners = []            # parameter -> the buckets it touches
        self._param_starts = []      # parameter -> its first element in the flat buffer
        self._inplace = set()        # parameters whose gradient (also) arrives in place, past autograd (see the class docstring)
        self._profile = {}           # such a parameter -> (in-place writes, hook firings) of one armed backward, once learned
        self._seen = []
        self._done = []
        self._ptr_param = {}         # address of a .grad region handed to an op -> parameter (memo of _param_of)
        self._listener = None
        if self.overlap:
            off = 0
            starts = [lo for lo, _ in self.buckets]
            import bisect

            for i, p in enumerate(self.params):
                # a parameter belongs to the LAST bucket it touches: that bucket waits for it, earlier ones it straddles do not
                # need to (they are issued after it, buckets go last -> first)... a straddling parameter is therefore counted in
                # BOTH buckets it touches
                first = bisect.bisect_right(starts, off) - 1
                last = bisect.bisect_right(starts, off + p.numel() - 1) - 1
                owners = list(range(first, last + 1))
                for b in owners:
                    self._bucket_params[b] += 1
                self._owners.append(owners)
                self._param_starts.append(off)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
                off += p.numel()
            from . import _lib as L

            self._listener = _GradSinkListener(self)
            L.GRAD_SINK_LISTENERS.append(self._listener)
        self.flat_param = None
        if self.sharded:
            self._flatten_parameters(pad)
        if world_size > 1:
            self.broadcast_parameters()

    # ------------------------------------------------------------------ layout
    def _flatten_parameters(self, pad: int):
        """parameters as views of one flat buffer too (what all_gather_params moves); models that already keep their
        parameters flat (HalfUNetMI355X) are left alone when the layout matches."""
        flat = torch.zeros(self.total + pad, dtype=torch.float32, device=self.flat_grad.device)
        off = 0
        for p in self.params:
            n = p.numel()
            flat[off : off + n].copy_(p.data.reshape(-1))
            p.data = flat[off : off + n].view_as(p)
            off += n
        self.flat_param = flat

    def shards(self):
        """This rank's [lo, hi) ranges, one per bucket (equal split of every bucket)."""
        out = []
        for lo, hi in self.buckets:
            n = (hi - lo) // self.world_size
            out.append((lo + self.rank * n, lo + (self.rank + 1) * n))
        return out

    def broadcast_parameters(self, src: int = 0):
        """Same initial weights on every rank (DDP's constructor does the same)."""
        from . import _lib as L

        L.invalidate_param_caches()   # (written through .data: tensor._version does not move)
        if self.flat_param is not None:
            dist.broadcast(self.flat_param, src)
            return
        for p in self.params:
            dist.broadcast(p.data, src)

    def zero_grad(self):
        """One fill over the flat buffer instead of one per parameter (``optimizer.zero_grad(set_to_none=False)``)."""
        self.wait()
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

    # ------------------------------------------------------------------ exchange overlapped with backward
    def arm(self):
        """Call before the backward of the micro-batch whose gradients will be exchanged (the last one of an accumulation window):
        from now on a bucket goes onto the communication stream the moment the gradients of all its parameters are final."""
        if not self.overlap:
            return
        self.wait()
        if not self._grads_are_views():
            # a grad tensor was replaced since the last exchange (optimizer.zero_grad(set_to_none=True), module.zero_grad(), a
            # `p.grad = ...` assignment): a bucket issued from a hook would reduce the flat buffer's stale bytes while the real
            # gradient sits elsewhere.  Stay un-armed: all_reduce_grads() then regathers and runs the ordinary exchange.
            self._armed = False
            self.issued_in_backward = 0
            return
        self._armed = True
        self._issued = [False] * len(self.buckets)
        self._pending = list(self._bucket_params)
        self._seen = [[0, 0] for _ in self.params]
        self._done = [False] * len(self.params)
        self.issued_in_backward = 0

    def close(self):
        """Stop listening (hooks and in-place write reports); the exchange after the backward keeps working."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._listener is not None:
            self._listener.remove()
            self._listener = None
        self.overlap = False
        self._armed = False

    def _make_hook(self, idx):
        def hook(_param):
            self._event(idx, 1)
        return hook

    def _param_of(self, view) -> int:
        """The parameter whose slice of the flat gradient buffer ``view`` starts in (-1: not in this buffer)."""
        ptr = view.data_ptr()
        idx = self._ptr_param.get(ptr)
        if idx is None:
            import bisect

            off = ptr - self.flat_grad.data_ptr()
            idx = -1
            if 0 <= off < 4 * self.total and view.device == self.flat_grad.device:
                idx = bisect.bisect_right(self._param_starts, off // 4) - 1
            self._ptr_param[ptr] = idx
        return idx

    def _sink_taken(self, view):
        idx = self._param_of(view)
        if idx >= 0:
            self._inplace.add(idx)

    def _written(self, views):
        if not self._armed:
            return
        for v in views:
            if v is not None:
                idx = self._param_of(v)
                if idx >= 0:
                    self._event(idx, 0)

    def _event(self, idx: int, kind: int):
        """kind 0: an in-place sum into parameter idx's gradient has been enqueued; kind 1: its post-accumulate hook fired."""
        if not self._armed:
            return
        seen = self._seen[idx]
        seen[kind] += 1
        if kind == 1 and idx not in self._inplace:
            self._finish(idx)            # everything goes through autograd: its sum over the AR steps IS the final gradient
            return
        self._inplace.add(idx)
        prof = self._profile.get(idx)
        if prof is None:
            return                       # learning pass for this parameter: its buckets stay back until all_reduce_grads
        if seen[0] > prof[0] or seen[1] > prof[1]:
            if any(self._issued[b] for b in self._owners[idx]):
                self._armed = False
                raise RuntimeError(
                    f"FlatDDP(overlap=True): parameter #{idx} received gradient contribution {seen} after its bucket had been issued "
                    f"on the {prof} contributions learned from an earlier step -- the number of uses of a parameter changed between "
                    "steps (a longer rollout, a branch taken for the first time).  Construct FlatDDP with overlap=False for such a "
                    "program, or keep the armed steps alike.")
            if self._done[idx]:          # not on the wire yet: take the parameter back and learn again from this pass
                self._done[idx] = False
                for b in self._owners[idx]:
                    self._pending[b] += 1
            del self._profile[idx]
            return
        if seen[0] == prof[0] and seen[1] == prof[1] and (prof[0] + prof[1]) > 0:
            self._finish(idx)

    def _finish(self, idx: int):
        if self._done[idx]:
            return
        self._done[idx] = True
        for b in self._owners[idx]:
            self._pending[b] -= 1
        self._issue_ready(from_hook=True)

    def _learn(self):
        """End of an armed backward: the counts of this pass become the expectation for the parameters that bypass autograd."""
        for idx in self._inplace:
            seen = tuple(self._seen[idx])
            if self._profile.get(idx) != seen:
                self._profile[idx] = seen

    def _issue_ready(self, from_hook: bool, force: bool = False):
        """Issue, last bucket first, every not yet issued bucket that is complete (or all of them with ``force``); stops at the
        first incomplete one so that the sequence of collectives is the same on every rank."""
        for b in range(len(self.buckets) - 1, -1, -1):
            if self._issued[b]:
                continue
            if not force and self._pending[b] > 0:
                break
            self._issued[b] = True
            if from_hook:
                self.issued_in_backward += 1
            self._on_comm_stream(lambda b=b: self._exchange_bucket(b))

    def _exchange_bucket(self, b: int):
        lo, hi = self.buckets[b]
        piece = self.flat_grad[lo:hi]
        inv = 1.0 / self.world_size
        if self.sharded:
            n = (hi - lo) // self.world_size
            mine = piece[self.rank * n : (self.rank + 1) * n]
            if dist.get_backend() == "gloo":   # gloo (CPU tests) has no reduce-scatter: same result.  (Chosen by the
                dist.all_reduce(piece, op=dist.ReduceOp.SUM)   # backend's name: a genuine RCCL failure must surface.)
            else:
                dist.reduce_scatter_tensor(mine, piece, op=dist.ReduceOp.SUM)
            mine.mul_(inv)
        else:
            dist.all_reduce(piece, op=dist.ReduceOp.SUM)
            piece.mul_(inv)

    # ------------------------------------------------------------------ exchange
    def _on_comm_stream(self, fn):
        """Run the collectives of ``fn`` on the communication stream, ordered after everything enqueued on the compute stream."""
        if self.comm_stream is None:
            fn()
            return
        self.comm_stream.wait_stream(torch.cuda.current_stream(self.flat_grad.device))
        with torch.cuda.stream(self.comm_stream):
            fn()
        self._inflight = True

    def wait(self):
        """Make the compute stream wait for the exchange in flight (no host synchronisation)."""
        if self._inflight and self.comm_stream is not None:
            torch.cuda.current_stream(self.flat_grad.device).wait_stream(self.comm_stream)
        self._inflight = False

    def all_reduce_grads(self, wait: bool = True):
        """Mean of the gradients over the ranks.  ``sharded``: only this rank's shards are complete afterwards (``shards()``)."""
        if self.world_size <= 1:
            return
        if not self._views_ok or not self._grads_are_views():
            if self._armed:
                # a gradient tensor was replaced DURING the armed backward: whatever the hooks issued reduced bytes that were not
                # the gradient.  Let those collectives finish, then exchange every bucket again from the regathered buffer (every
                # rank takes this branch together: it follows from the program, not from data).
                self.wait()
                if self.sharded and any(self._issued):
                    # a reduce-scatter already overwrote this rank's shard of those buckets with the MEAN and left the rest of the
                    # piece with local (RCCL) or summed (gloo) values: the parameters whose gradients still lived in the flat buffer
                    # cannot be exchanged a second time from it.  There is no copy to restore them from -- refuse
                    # loudly instead of stepping with a silently wrong gradient.
                    self._armed = False
                    raise RuntimeError(
                        "FlatDDP(sharded=True, overlap=True): a gradient tensor was replaced during an armed backward after "
                        f"{sum(self._issued)} bucket(s) had been reduce-scattered from the hooks; the flat buffer no longer holds the "
                        "local gradients of those buckets.  Keep p.grad in place during backward (no zero_grad(set_to_none=True) / "
                        "p.grad = ... inside it), or construct FlatDDP with overlap=False or sharded=False.")
                self._issued = [False] * len(self.buckets)
            self._regather()
        if self.overlap and self._armed:
            self._learn()
            # what the hooks have not issued yet (parameters that received no gradient in this pass never fire theirs)
            self._issue_ready(from_hook=False, force=True)
            self._armed = False
        else:
            def exchange():
                for b in range(len(self.buckets) - 1, -1, -1):       # last layers first
                    self._exchange_bucket(b)

            self._on_comm_stream(exchange)
        if wait:
            self.wait()
        if not self._views_ok:
            self.wait()
            self._scatter()

    def all_gather_params(self, wait: bool = True):
        """After a sharded optimizer step: every rank receives the other ranks' updated shards."""
        if not self.sharded:
            return

        def gather():
            for lo, hi in self.buckets:
                piece = self.flat_param[lo:hi]
                n = (hi - lo) // self.world_size
                mine = piece[self.rank * n : (self.rank + 1) * n].clone()
                if dist.get_backend() == "gloo":   # no flat form there
                    dist.all_gather([piece[r * n : (r + 1) * n] for r in range(self.world_size)], mine)
                else:
                    dist.all_gather_into_tensor(piece, mine)

        self._on_comm_stream(gather)
        from . import _lib as L

        L.invalidate_param_caches()   # (the other ranks' shards arrive through the flat buffer: tensor._version does not move)
        if wait:
            self.wait()

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


class GraphReplayMismatch(RuntimeError):
    """A captured training step whose replay does not reproduce the eager step (GraphedTrainingStep._verify)."""


AFFINITY_POLICY = [None]   # what pin_rank_to_cores decided for this process (bench.py: config.host_affinity_policy)


def _cpu_topology(cpus):
    """{cpu: (numa node, package, core id)} from /sys (Linux); None where that is not readable"""
    import glob
    import os

    node_of = {}
    for d in glob.glob("/sys/devices/system/node/node[0-9]*"):
        try:
            node = int(os.path.basename(d)[4:])
            for part in open(os.path.join(d, "cpulist")).read().strip().split(","):
                lo, _, hi = part.partition("-")
                for c in range(int(lo), int(hi or lo) + 1):
                    node_of[c] = node
        except (OSError, ValueError):
            return None
    topo = {}
    for c in cpus:
        try:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            topo[c] = (node_of.get(c, 0), int(open(base + "physical_package_id").read()), int(open(base + "core_id").read()))
        except (OSError, ValueError):
            return None
    return topo


def pin_rank_to_cores(local_rank: int, local_world: int):
    """Give each rank of a node its own set of the cores this process may run on (``os.sched_setaffinity``), BEFORE the rank's first
    GPU call: eight Python ranks, their autograd threads and RCCL's proxy threads otherwise wander over one another's cores (the
    reference's own 4-GPU runs lose 25-49 % per rank to the host, doc/num_steps.md:119-143).
    Topology-aware: the CPUs are grouped by NUMA node and PHYSICAL core (/sys/devices/system/cpu/*/topology) -- on an SMT
    host the ids are typically socket 0, socket 1, socket-0 siblings, socket-1 siblings, and contiguous slices of the id list would hand
    rank r and rank r + N/2 the two hyper-threads of the same cores; a rank gets whole physical cores (all their hardware threads),
    ranks are dealt over the NUMA nodes in blocks (ranks 0..N/k-1 on node 0, ...), which is also how the GPUs of a node are usually
    attached.  Nothing is changed when the launcher has already bound this rank (the inherited affinity is narrower than the machine:
    numactl / torchrun binding), with one rank, with fewer physical cores than ranks, or with P4C_NO_AFFINITY=1.
    Returns the sorted CPU list of this rank or None; AFFINITY_POLICY[0] names what was decided."""
    import os

    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        AFFINITY_POLICY[0] = "unchanged (one rank per host or no affinity interface)"
        return None
    if os.environ.get("P4C_NO_AFFINITY") == "1":
        AFFINITY_POLICY[0] = "unchanged (P4C_NO_AFFINITY=1)"
        return None
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < (os.cpu_count() or len(allowed)):
        AFFINITY_POLICY[0] = f"unchanged (the launcher bound this rank to {len(allowed)} of {os.cpu_count()} CPUs)"
        return None
    topo = _cpu_topology(allowed)
    if topo is None:
        per = len(allowed) // local_world
        if per < 1:
            AFFINITY_POLICY[0] = "unchanged (fewer CPUs than ranks)"
            return None
        mine = allowed[local_rank * per:(local_rank + 1) * per]
        AFFINITY_POLICY[0] = f"contiguous slice of the CPU ids ({per} per rank; topology not readable)"
    else:
        cores = {}      # (node, package, core) -> hardware threads
        for c, key in topo.items():
            cores.setdefault(key, []).append(c)
        ordered = sorted(cores)                      # by NUMA node, then package, then core
        per = len(ordered) // local_world
        if per < 1:
            AFFINITY_POLICY[0] = "unchanged (fewer physical cores than ranks)"
            return None
        mine = sorted(c for key in ordered[local_rank * per:(local_rank + 1) * per] for c in cores[key])
        nodes = sorted({topo[c][0] for c in mine})
        AFFINITY_POLICY[0] = (f"{per} physical cores per rank ({len(mine)} hardware threads), whole cores, ranks in blocks over the NUMA "
                              f"nodes; this rank: node(s) {nodes}")
    os.sched_setaffinity(0, mine)
    torch.set_num_threads(max(1, min(len(mine), torch.get_num_threads())))
    return mine


class RolloutParamProxies:
    """Per-AR-step stand-ins for a module's parameters.  The T model calls of a rollout share their parameters, so autograd adds T
    gradient contributions per parameter one kernel at a time (``AccumulateGrad``: 478 parameters x 6 steps = 2 400 tiny launches per
    UNetRPP training step, 4 % of its kernel time).  Here call t runs on detached leaf views of the parameters (same storage, their own
    ``.grad``: the first contribution is kept by reference, no kernel) and one callback at the end of the backward adds the T gradient
    sets into ``param.grad`` with multi-tensor launches (``torch._foreach_add_``: a few dozen launches).  Same sums in a different
    order of additions.  Single-process only: a gradient exchange driven by per-parameter hooks (FlatDDP, N > 1) never sees these.

    Contract: the transfer runs as an autograd-engine callback of every backward that reaches the tensor given to ``attach`` (any
    number of backwards: ``retain_graph``, several losses); gradients a backward left in the stand-ins WITHOUT reaching that tensor
    are transferred by the next ``begin()`` -- never dropped."""

    def __init__(self, model: torch.nn.Module):
        self.model = model
        self.named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.sets = []        # reused from step to step: [{name: stand-in}]
        self.used = 0
        self.dirty = False    # stand-ins were handed out since the last transfer (their .grad may hold something)

    def begin(self):
        if self.dirty:
            # a backward that never reached the attached tensor left its gradients in the stand-ins: keep them -- they belong to
            # the PREVIOUS step and arrive after its zero_grad (a host that wants them in time calls finalize() itself)
            import warnings

            warnings.warn("RolloutParamProxies: gradients of the previous rollout were still in the stand-ins (its backward did not "
                          "reach the attached tensor); they are added to .grad now")
            self.finalize()
        self.used = 0

    def call(self, x):
        if self.used == len(self.sets):
            self.sets.append({})
        prox = self.sets[self.used]
        self.used += 1
        self.dirty = True
        for n, p in self.named:
            q = prox.get(n)
            if q is None or q.data_ptr() != p.data_ptr() or q.shape != p.shape:
                q = p.detach().requires_grad_(True)
                q._p4c_owner = p      # caches keyed on the owner of a weight's storage (weight images, casts) see the parameter
                prox[n] = q
        return torch.func.functional_call(self.model, prox, (x,))

    def attach(self, t: torch.Tensor):
        """``t``: a tensor of the rollout's result that every backward through the rollout reaches."""
        if self.used and t.requires_grad:
            t.register_hook(self._on_backward)

    def _on_backward(self, g):
        torch.autograd.Variable._execution_engine.queue_callback(self.finalize)
        return None

    def finalize(self):
        self.dirty = False
        with torch.no_grad():
            for prox in self.sets[: self.used]:
                tgt, src = [], []
                for n, p in self.named:
                    q = prox[n]
                    g = q.grad
                    if g is None:
                        continue
                    q.grad = None
                    if p.grad is None:
                        p.grad = g.clone()
                    elif g.dtype == p.grad.dtype and g.shape == p.grad.shape and g.is_contiguous() and p.grad.is_contiguous():
                        tgt.append(p.grad)
                        src.append(g)
                    else:
                        # kept out of the multi-tensor call: ONE pair off its fast route (another dtype, a strided view) sends the
                        # whole list down torch's per-tensor fallback -- one launch per parameter and AR step again
                        p.grad.add_(g)
                if tgt:
                    torch._foreach_add_(tgt, src)
        # (`used` stays: a second backward through the same rollout finds its stand-ins here again; begin() resets it)


class GraphedTrainingStep:
    """One micro-batch -- ``training_step`` (rollout + loss) and its ``backward`` -- captured in a HIP graph and replayed.

    The eager step of a small-kernel model is bound by the host, not the GPU: a hierarchical GNN issues ~10^4 launches per step
    from Python (HiLAM at 512x512: 164 ms eager per step for ~30 ms of kernels).  Every entry point of the C ABI only enqueues on
    the caller's stream and takes its work-spaces from the caller, so a whole step is capturable; replaying it costs one launch.
    Inputs live in static tensors (a replay copies the new batch in), gradients accumulate into the parameters' existing ``.grad``
    buffers (FlatDDP's flat bucket), so ``all_reduce_grads`` / ``optimizer.step`` / ``zero_grad`` run outside the graph as usual.
    Shapes must not change between steps; a module whose step synchronises with the host cannot be captured (the constructor
    raises, nothing is left half-captured)."""

    def __init__(self, module, sample_batch, loss_scale: float = 1.0, warmup: int = 3, verify: bool = True):
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

        verify = verify and warmup >= 3
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        # A step that draws random numbers (UNETR++'s published block: the Dropout2d in front of conv8) is compared on EQUAL draws:
        # the device generator is re-seeded before every eager reference pass and before every checking replay (a captured kernel
        # reads seed and offset from the generator's state at replay time, and the step consumes them in the same order either way),
        # so the check stays the strict "replay == eager" one instead of a statistical one over two independent draws -- which accepted
        # or rejected the same correct capture from run to run (round 6).  The caller's generator state is put back afterwards.
        dev = next(module.parameters()).device
        rng_state = torch.cuda.get_rng_state(dev)
        self._reseed = lambda: torch.cuda.manual_seed(0x5EED)

        def grads():   # every gradient of the module as one fp32 vector (None until the first backward has allocated them)
            if any(p.grad is None for _, p in named):
                return None
            return torch.cat([p.grad.detach().reshape(-1).float() for _, p in named])

        snaps, eager_loss = [], None
        # the warm-up / capture / verification passes are not training steps: the module's per-step loss list is cut back
        # to its length on entry when the constructor is done
        self._loss_log_len = len(getattr(module, "training_step_losses", []) or [])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(warmup):      # lazy initialisation (edge sets, kernel attributes, allocator pools) happens here
                self._reseed()
                eager_loss = run(i)
                if verify and i >= warmup - 3:
                    snaps.append(grads())
        torch.cuda.current_stream().wait_stream(side)
        from . import _lib as L

        # keep_graph: the captured hipGraph_t stays accessible until it is instantiated below
        try:
            self.graph, rewritable = torch.cuda.CUDAGraph(keep_graph=True), True
        except TypeError:   # a torch without keep_graph: no access to the captured graph, the replay check below is the only guard
            self.graph, rewritable = torch.cuda.CUDAGraph(), False
        L.CAPTURE_SCOPE[0] = {}
        try:
            with torch.cuda.graph(self.graph):
                self.loss = run(warmup)
        finally:
            L.CAPTURE_SCOPE[0] = None
        # On this stack a memset node of a HIP graph writes the right bytes in the first replay only, and torch's
        # multi-block reductions -- every broadcast-added bias gradient, for one -- zero their semaphores with one: rewrite the
        # memset nodes as fill-kernel nodes before the graph is instantiated (csrc/graphfix.hip)
        self.memset_nodes = None
        if rewritable:
            import ctypes

            replaced, left = ctypes.c_int(), ctypes.c_int()
            L.check(L.lib().p4c_graph_replace_memsets(ctypes.c_void_p(self.graph.raw_cuda_graph()), ctypes.byref(replaced), ctypes.byref(left)),
                    "p4c_graph_replace_memsets")
            self.memset_nodes = (replaced.value, left.value)
            self.graph.instantiate()
        self.warmup_backwards = warmup + 1   # gradient contributions already accumulated by construction: zero_grad() after
        self.verified = None
        try:
            if verify and all(g is not None for g in snaps):
                self._verify(named, snaps, eager_loss.clone(), grads, run)
        finally:
            torch.cuda.set_rng_state(rng_state, dev)
            log = getattr(module, "training_step_losses", None)
            if isinstance(log, list):
                del log[self._loss_log_len:]

    def _verify(self, named, snaps, eager_loss, grads, run):
        """Replay the captured step once on the captured batch and hold its gradient contribution and loss against the eager
        warm-up passes on the same batch and weights.  A replay is only as good as every library call inside it is capture-safe,
        and that is not ours to promise: at the 512 x 512 sizes two library routes returned garbage / NaN from replays while
        every eager step was fine (the column-sum reduction behind Linear bias gradients, and 1x1 convolutions as bias-epilogue
        GEMMs -- DESIGN.md 7b).  So the graph has to earn its use: per parameter, the replay's gradient must match the eager one to
        the level two eager passes match each other (steps with a random element -- masks, dropout -- differ between eager
        passes too and are not checked), else the constructor raises GraphReplayMismatch and callers stay eager.
        Both failures showed only AFTER an optimizer step (a replay that keeps reading something derived from the parameters at
        capture time), so the comparison is made twice: on the captured weights, and again after every parameter has been scaled by
        1.25 in place (restored afterwards; large enough that a replay reading capture-time weights is far outside any tolerance)."""
        g1, g2, g3 = snaps
        inc_a, inc_b = g2 - g1, g3 - g2          # two eager contributions
        self._reseed()
        self.graph.replay()
        self.warmup_backwards += 1
        inc_g = grads() - g3
        graph_loss = self.loss.float().clone()
        lengths = torch.tensor([p.numel() for _, p in named], device=inc_g.device)
        self._compare(named, lengths, inc_a, inc_b, inc_g, eager_loss, graph_loss, "")
        first = self.verified
        # ---- the same with changed parameters
        from . import _lib as L

        params = [p for _, p in named]
        keep = [p.detach().clone() for p in params]
        try:
            with torch.no_grad():
                torch._foreach_mul_(params, 1.25)
            L.PARAM_EPOCH[0] += 1
            g4 = grads()
            self._reseed()
            eager_loss2 = run(-1).float().clone()
            g5 = grads()
            self._reseed()
            self.graph.replay()
            self.warmup_backwards += 2
            g6 = grads()
            graph_loss2 = self.loss.float().clone()
            # eager-vs-eager spread: the one measured on the captured weights
            self._compare(named, lengths, inc_a, g5 - g4, g6 - g5, eager_loss2, graph_loss2, " after a parameter update", spread_ref=inc_b)
        finally:
            with torch.no_grad():
                for p, k in zip(params, keep):
                    p.copy_(k)
            L.PARAM_EPOCH[0] += 1
        self.verified = first + "; also after a parameter update"

    def _compare(self, named, lengths, inc_a, inc_b, inc_g, eager_loss, graph_loss, when, spread_ref=None):
        """inc_b: the eager contribution the replay's (inc_g) is held against; inc_a (with spread_ref or inc_b): two eager contributions
        on one set of weights, whose difference is the step's own spread."""

        def per_param(diff, ref):   # relative L2 error of every parameter's gradient
            num = torch.segment_reduce(diff.double().square(), "sum", lengths=lengths)
            den = torch.segment_reduce(ref.double().square(), "sum", lengths=lengths)
            return (num / den.clamp_min(1e-300)).sqrt(), den

        ref2 = inc_b if spread_ref is None else spread_ref
        base, _ = per_param(ref2 - inc_a, ref2)
        got, den = per_param(inc_g - inc_b, inc_b)
        live = den > 1e-24 * den.sum()     # a parameter whose gradient is (numerically) nothing has no relative error to judge
        finite = bool(torch.isfinite(inc_g).all()) or not bool(torch.isfinite(inc_b).all())
        # two eager passes on one batch and one set of weights that differ by percents: a random element in the step (masks,
        # dropout), or bf16 activations + atomics whose accumulation order moves LeakyReLU / softmax branches (SwinUNetR and
        # UNetRPP at 512 x 512).  Such a step is held to its own noise: only a replay far outside it -- or not finite -- fails
        # (any step that is not reproducible to ~1e-5 counts: classifying by "many parameters differ by percents" put UNetRPP on
        # either side from one run to the next, and the strict bounds then rejected a correct capture)
        noisy = bool(base[live].median() > 1e-5) if bool(live.any()) else False
        if noisy:
            # bounded by the MEASURED noise, per parameter and overall (a replay that adds nothing to a gradient has error 1.0 and
            # one that reads stale weights a few percent: both must fail for a step whose own spread is below that)
            med = float(base[live].median())
            tol = torch.maximum(torch.maximum(10.0 * base, torch.full_like(base, 3.0 * med)), torch.full_like(base, 5e-2))
        else:
            # (a reproducible step -- every model of the registry since the library convolutions are pinned to deterministic solvers,
            # ops_model.library_conv2d -- replays the SAME kernels on the same data: measured worst relative error 2.4e-7)
            tol = torch.maximum(10.0 * base, torch.full_like(base, 1e-3))
        bad = (live & ~(got <= tol)).nonzero().flatten().tolist()      # `~(<=)`: NaN counts as bad
        # the loss is far less noisy than per-parameter gradients (it moves in the 6th digit where gradients move by percents);
        # the whole gradient's norm and direction are checked too
        gn, en = float(inc_g.double().norm()), float(inc_b.double().norm())
        cosv = float(torch.dot(inc_g.double(), inc_b.double()) / max(gn * en, 1e-300))
        spread = float((ref2 - inc_a).double().norm() / max(float(ref2.double().norm()), 1e-300))
        # (a step with a random element -- a mask draw, dropout -- moves its loss as much as its gradients between two eager
        # passes: the loss bound follows the measured spread there, 0.5 % otherwise)
        loss_tol = (max(5e-3, 3.0 * spread) if noisy else 1e-4) * abs(float(eager_loss)) + 1e-6
        loss_ok = bool(torch.isfinite(graph_loss)) and abs(float(graph_loss) - float(eager_loss)) <= loss_tol
        whole_ok = finite and abs(gn - en) <= max(10.0 * spread, 2e-2 if noisy else 1e-3) * en and cosv >= 1.0 - max(50.0 * spread * spread, 1e-3 if noisy else 1e-5)
        if not whole_ok:
            loss_ok = False
        if bad or not finite or not loss_ok:
            worst = max(bad, key=lambda i: float(torch.nan_to_num(got[i], nan=float("inf")))) if bad else None
            raise GraphReplayMismatch(
                f"the replayed training step does not reproduce the eager one{when}: "
                + (f"{len(bad)} of {len(named)} parameter gradients differ, worst {named[worst][0]} "
                   f"(relative error {float(got[worst]):.3g}, eager-vs-eager {float(base[worst]):.3g}); " if bad else "")
                + f"loss eager {float(eager_loss):.6g} vs replay {float(graph_loss):.6g}; gradient norm eager {en:.6g} vs replay {gn:.6g}, "
                  f"cosine {cosv:.6f} (eager-vs-eager spread {spread:.3g})")
        worst_got = float(got[live].max()) if bool(live.any()) else 0.0
        if noisy:
            self.verified = (f"replay within the step's own eager-vs-eager spread on {len(named)} parameter gradients (non-deterministic step: "
                             f"median eager-vs-eager {float(base[live].median()):.2g}, median replay-vs-eager {float(got[live].median()):.2g})")
        else:
            self.verified = f"replay == eager on {len(named)} parameter gradients (worst relative error {worst_got:.2g})"

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
        loss = self.loss.detach().clone()   # (self.loss is ONE static tensor: every replay overwrites it)
        log = getattr(self.module, "training_step_losses", None)
        if isinstance(log, list):           # what training_step itself does on an eager step (lightning.py:516 here)
            log.append(loss)
        return loss



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
        ddp = FlatDDP(module, overlap=True)   # (several buckets and several ranks only: otherwise one exchange after the backward)
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
                    import warnings

                    host_random = float(getattr(module, "mask_ratio", 0) or 0) != 0 or any(
                        isinstance(m_, torch.nn.modules.dropout._DropoutNd) and m_.p > 0 for m_ in module.modules())
                    # (the warm-up passes draw random numbers too: the training run's generators continue where they were)
                    rng_cpu = torch.get_rng_state()
                    rng_dev = torch.cuda.get_rng_state(self.device) if self.device.type == "cuda" else None
                    try:
                        if host_random:
                            # a host-side random element (the block mask is drawn on the CPU and copied, lightning.py:580-581;
                            # dropout) would be frozen to one draw by a capture -- and the blocking copy is illegal inside one
                            raise GraphReplayMismatch("the step has a host-side random element (mask_ratio / dropout)")
                        graphed = GraphedTrainingStep(module, batch, loss_scale=1.0 / self.accumulate_grad_batches)
                    except GraphReplayMismatch as exc:   # a replay that is not the eager step is not used: stay eager, say so
                        warnings.warn(f"HIP-graph replay rejected, training continues with eager launches: {exc}")
                        use_graph = False
                    except torch.cuda.OutOfMemoryError:
                        raise   # not a property of the capture: the eager step would meet the same wall, far from its cause
                    except RuntimeError as exc:
                        # Only what says "this step cannot be captured" turns into eager launches: an operation that is not
                        # permitted while a stream is capturing (a synchronising call, a blocking copy, an allocation outside the
                        # graph's pool).  Anything else -- an illegal address, a failed launch, a library error -- is a device
                        # fault and must surface here.
                        msg = str(exc).lower()
                        if not any(k in msg for k in ("captur", "graph", "operation not permitted", "streamcapture")):
                            raise
                        warnings.warn(f"HIP-graph capture failed ({type(exc).__name__}: {exc}); training continues with eager launches")
                        use_graph, graphed = False, None
                    finally:
                        ddp.flat_grad.copy_(saved)
                        for b, keep in buffers:
                            b.copy_(keep)
                        torch.set_rng_state(rng_cpu)
                        if rng_dev is not None:
                            torch.cuda.set_rng_state(rng_dev, self.device)
                if graphed is not None and graphed.accepts(batch):
                    loss = graphed(batch)
                else:   # eager; also a batch whose shape differs from the captured one (a short last batch)
                    loss = module.training_step(batch, i)
                    if pending + 1 == self.accumulate_grad_batches or last_of_epoch:
                        ddp.arm()   # the micro-batch that steps: buckets go out while its backward is still running
                    (loss / self.accumulate_grad_batches).backward()
                self.train_step_losses.append(loss.detach().clone() if graphed is not None else loss.detach())
                pending += 1
                # Lightning steps every accumulate_grad_batches micro-batches AND on the last batch of an epoch: leftovers do not
                # leak into the next epoch.  Non-stepping micro-batches do not sync
                if pending == self.accumulate_grad_batches or last_of_epoch:
                    pending = 0
                    ddp.all_reduce_grads()
                    if ddp.sharded and hasattr(opt, "step_shards"):
                        opt.step_shards(ddp.shards())
                        ddp.all_gather_params()
                    else:
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
