"""Launch grouping for the mesh GNNs (include/py4cast_hip.h: p4c_node_proj_fwd / _dgrad / _wgrad, p4c_grad_reduce_defer / _flush;
csrc/nodeproj.hip).

GraphLAM / HiLAM / HiLAMParallel (config/CLI/model/graphlam.yaml:19-26, hilam.yaml, hilamparallel.yaml; taken from mfai at
py4cast/models.py:66-89) apply ~190 InteractionNets per optimizer step, most of them on mesh levels of a few dozen to a few thousand
rows, where a launch costs its latency whatever it computes.  Two things cut the number of dependent launches:

* ``node_proj``: the 64 x 64 blocks of the distributed first Linears that multiply one node tensor (sender / receiver part of the edge
  MLP, receiver part of the node-update MLP) as ONE launch forward, one for the data gradient (K = 64 n) and one for the weight
  gradients -- instead of n row-GEMMs each way plus, per block, a weight-gradient product, its reduction and a ``+=``;
* ``GradQueue``: the reductions of the parameter-gradient partials (this module's and ops_mlp.row_mlp's) are queued during a backward
  pass and run 32 jobs per launch when the autograd engine finishes the pass (``queue_callback``), in submission order: bit-identical
  to reducing after every call, ~550 dependent 5 us launches fewer per HiLAM step.

No CPU fallback.
"""

import ctypes
from typing import List, Sequence

import torch

from . import _lib as L
from .ops_rows import grad_view, row_linear

_P3 = ctypes.c_void_p * 3
_I3 = ctypes.c_int32 * 3


class GradQueue:
    """Deferred reduction of the gradient partials of ONE backward pass.  ``begin(keep)`` (from inside an autograd Function's backward)
    switches the library to queueing, registers the flush with the engine once per pass and keeps ``keep`` (the partials' workspace)
    alive until the flush has been enqueued.  Outside a backward pass (no graph task) nothing is deferred."""

    enabled = True          # False: every call reduces at once (the A/B reference of tests/test_nodeproj_gpu.py)
    _task = -1
    _keep: List[torch.Tensor] = []
    _views: List[torch.Tensor] = []      # the .grad regions the queued reductions will add into (told to L.grad_written at the flush)
    _stream = None

    @classmethod
    def begin(cls, keep: torch.Tensor) -> None:
        if not cls.enabled:
            return
        task = torch._C._current_graph_task_id()
        if task < 0:
            return
        if cls._task != task:
            if cls._task >= 0:      # a pass that died before its callback ran: its queued jobs are void
                L.lib().p4c_grad_reduce_defer(-1)
                cls._keep, cls._views = [], []
            cls._task = task
            cls._stream = torch.cuda.current_stream(keep.device)
            L.lib().p4c_grad_reduce_defer(1)
            torch.autograd.Variable._execution_engine.queue_callback(cls.flush)
        cls._keep.append(keep)

    @classmethod
    def active(cls) -> bool:
        """the reductions of the backward pass in progress are being queued"""
        return cls.enabled and cls._task >= 0 and cls._task == torch._C._current_graph_task_id()

    @classmethod
    def wrote(cls, views) -> None:
        """After the launch that produced the partials: ``views`` (regions of parameters' .grad) receive their sums when the queue is
        flushed -- or have just received them when nothing is being deferred."""
        if not L.GRAD_SINK_LISTENERS:
            return
        views = [v for v in views if v is not None]
        if cls._task >= 0 and cls._task == torch._C._current_graph_task_id():
            cls._views.extend(views)
        else:
            L.grad_written(*views)

    @classmethod
    def flush(cls) -> None:
        if cls._task < 0:
            return
        views = cls._views
        try:
            L.check(L.lib().p4c_grad_reduce_flush(ctypes.c_void_p(cls._stream.cuda_stream)), "p4c_grad_reduce_flush")
        finally:
            L.lib().p4c_grad_reduce_defer(0)
            cls._task, cls._keep, cls._views, cls._stream = -1, [], [], None
        if views:
            L.grad_written(*views)


def _arr(tensors: Sequence[torch.Tensor]):
    return _P3(*[t.data_ptr() for t in tensors], *([None] * (3 - len(tensors))))


def _lds(tensors: Sequence[torch.Tensor]):
    return _I3(*[t.stride(0) for t in tensors], *([64] * (3 - len(tensors))))


class _NodeProj(torch.autograd.Function):
    """(x W_1^T, ..., x W_n^T) for bf16 node rows x (R, 64) and 64 x 64 blocks W_i of fp32 master weights, as one autograd node whose
    backward ADDS every weight gradient into ``gws[i]`` -- the view of the parameter's ``.grad`` that corresponds to W_i.
    ``passthrough``: x itself is returned as one more output (the residual operand of the node-update MLP): its gradient then arrives
    HERE and is added inside the data-gradient launch instead of by an element-wise launch of autograd's."""

    @staticmethod
    def forward(ctx, x, n, passthrough, *args):
        ws, gws = args[:n], args[n:]
        xc = x.contiguous()
        R = xc.shape[0]
        ys = [torch.empty(R, 64, dtype=xc.dtype, device=xc.device) for _ in range(n)]
        wd = [w.detach() for w in ws]
        L.call("p4c_node_proj_fwd", L.ptr(xc), R, n, _arr(wd), _lds(wd), _arr(ys), L.stream(xc.device),
               alg_bytes=R * 128 * (1 + n) + n * 64 * 64 * 4, alg_flops=2 * R * 64 * 64 * n)
        ctx.save_for_backward(xc, *wd)
        ctx.gws, ctx.n, ctx.passthrough = gws, n, passthrough
        return tuple(ys) + ((x,) if passthrough else ())

    @staticmethod
    def backward(ctx, *dys):
        x, *wd = ctx.saved_tensors
        n, R = ctx.n, x.shape[0]
        none = (None,) * (2 + 2 * n)
        dres = dys[n].contiguous() if (ctx.passthrough and dys[n] is not None) else None
        live = [i for i in range(n) if dys[i] is not None]
        if not live:
            return (dres,) + none
        dyl = [dys[i].contiguous() for i in live]
        wl = [wd[i] for i in live]
        m = len(live)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            L.call("p4c_node_proj_dgrad", _arr(dyl), R, m, _arr(wl), _lds(wl), L.ptr(dx), L.ptr(dres), L.stream(x.device),
                   alg_bytes=R * 128 * (1 + m + (dres is not None)) + m * 64 * 64 * 4, alg_flops=2 * R * 64 * 64 * m)
        gl = [ctx.gws[i] for i in live]
        ws = torch.empty(max(L.lib().p4c_node_proj_wgrad_workspace_bytes(R, m) // 4, 1), dtype=torch.float32, device=x.device)
        GradQueue.begin(ws)
        L.call("p4c_node_proj_wgrad", _arr(dyl), L.ptr(x), R, m, _arr(gl), _lds(gl), L.ptr(ws), L.stream(x.device),
               alg_bytes=R * 128 * (1 + m) + ws.numel() * 4, alg_flops=2 * R * 64 * 64 * m)
        GradQueue.wrote(gl)
        return (dx,) + none


def node_proj_ok(x: torch.Tensor, weights) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 64 and x.shape[0] >= 1 and 1 <= len(weights) <= 3
            and all(w.dtype == torch.float32 and w.dim() == 2 and tuple(w.shape) == (64, 64) and w.stride(1) == 1 for w in weights))


def node_proj(x: torch.Tensor, weights, grads_in_place: bool = True, passthrough: bool = False):
    """``[x @ w.T for w in weights]``: bf16 rows of 64 features, up to three 64 x 64 fp32 blocks (column slices of wider Linear weights
    are fine).  With gradient buffers on every weight (FlatDDP / Trainer allocate them) ONE launch per direction (module docstring);
    otherwise one ``ops_rows.row_linear`` each (autograd-returned weight gradients).  ``passthrough``: x is appended to the results --
    use THAT tensor as the residual of the node update that follows, and its gradient is summed inside the data-gradient launch."""
    L.require_cuda(x)
    weights = list(weights)
    if grads_in_place and node_proj_ok(x, weights) and torch.is_grad_enabled() and all(w.requires_grad for w in weights):
        gws = [grad_view(w) for w in weights]
        if all(g is not None and g is not False and g.stride(1) == 1 for g in gws):
            return _NodeProj.apply(x, len(weights), bool(passthrough), *weights, *gws)
    return tuple(row_linear(x, w, grads_in_place=grads_in_place) for w in weights) + ((x,) if passthrough else ())
