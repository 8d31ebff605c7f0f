"""Autograd wrapper of the fused row MLP (include/py4cast_hip.h: p4c_row_mlp_fwd / p4c_row_mlp_bwd).

``y = LayerNorm(SiLU(x W1^T + b1 [+ ga[ia] + gb[ib]]) W2^T + b2) [+ res]`` -- the shape of every MLP of GraphLam / HiLAM
(config/CLI/model/graphlam.yaml:21-22) -- as ONE kernel each way over bf16 rows; on edges the gathered addends are the sender /
receiver parts of the distributed first layer (py4cast_amd.ops_graph).  No CPU fallback.
"""

import ctypes
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib as L
from ._lib_model import RowMlpDesc, RowMlpGradSinks
from .ops_graph import EdgeSet, _segment_sum_pair_raw, _segment_sum_raw

MAX_K = 80


def supported(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and w1.shape[0] == 64 and w2.shape[1] == 64 and w2.shape[0] <= 64
            and w1.shape[1] <= MAX_K)


def _desc(x, K, w1, b1, w2, b2, gamma, beta, eps, ga, ia, gb, ib, res, out, out_res, dy=None, dy_res=None, dx=None, dpre=None):
    p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    return RowMlpDesc(rows=x.shape[0], x=p(x), k=K, k_real=w1.shape[1], w1=p(w1), ldw1=w1.stride(0), b1=p(b1), w2=p(w2), b2=p(b2),
                      o_real=w2.shape[0], gamma=p(gamma), beta=p(beta), eps=eps, gather_a=p(ga), index_a=p(ia), gather_b=p(gb),
                      index_b=p(ib), res=p(res), out=p(out), out_res=p(out_res), dy=p(dy), dy_res=p(dy_res), dx=p(dx), dpre=p(dpre))


_PREPARED = {}   # (parameter addresses, shapes) -> (parameter versions, blob)


def _prepared(d: RowMlpDesc, K: int, tensors, device, owners=None) -> torch.Tensor:
    """The parameters re-laid into the kernels' operand images (p4c_row_mlp_prepare): once per parameter version in eager mode;
    under HIP-graph capture re-issued (once per capture when the capturing code opened a scope, _lib.CAPTURE_SCOPE), so that a
    replay re-lays the CURRENT parameters."""
    capturing = torch.cuda.is_current_stream_capturing()
    owners = tensors if owners is None else owners   # the caller's parameter objects (`tensors` may be temporaries over their storage)
    key = (K, d.k_real, d.ldw1, d.o_real, d.eps) + tuple(None if t is None else t.data_ptr() for t in tensors)
    ver = (L.PARAM_EPOCH[0],) + tuple(None if t is None else t._version for t in tensors)
    scope = L.capture_cache()   # capture with an open scope: once per capture (the AR steps and the backward share the images)
    cache = _PREPARED if not capturing else scope
    if cache is not None:
        hit = cache.get(("mlp",) + key)
        if hit is not None and hit[0] == ver and L.owners_alive(hit[2], owners):
            return hit[1]
    blob = torch.empty(L.lib().p4c_row_mlp_prepared_bytes(K), dtype=torch.uint8, device=device)
    L.call("p4c_row_mlp_prepare", ctypes.byref(d), L.ptr(blob), L.stream(device))
    if cache is not None:
        cache[("mlp",) + key] = (ver, blob, L.owner_refs(owners))
    return blob


class _RowMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, ga, gb, res, edges: Optional[EdgeSet], eps: float, want_out: bool, sinks=None,
                owners=None, res_is_x: bool = False):
        R, K = x.shape
        ctx.res_is_x = bool(res_is_x) and K == 64
        x = x.contiguous()
        for t in (w1, w2):
            if t.dtype != torch.float32 or t.stride(1) != 1:
                raise L.P4CError("row_mlp: weights must be fp32 with unit column stride")
        w2c = w2.contiguous()
        f32 = lambda t: None if t is None else t.detach().float().contiguous()  # noqa: E731
        b1c, b2c, gc, bc = f32(b1), f32(b2), f32(gamma), f32(beta)
        gac = None if ga is None else ga.contiguous()
        gbc = None if gb is None else gb.contiguous()
        resc = None if res is None else res.contiguous()
        out = torch.empty(R, 64, dtype=x.dtype, device=x.device) if want_out else None
        out_res = torch.empty(R, 64, dtype=x.dtype, device=x.device) if res is not None else None
        ia = edges.src if (ga is not None and edges is not None) else None   # no edge set: ga is a row-aligned addend
        ib = edges.dst if (gb is not None and edges is not None) else None
        if edges is None and any(t is not None and t.shape[0] != R for t in (gac, gbc)):
            raise L.P4CError("row_mlp: without an edge set the addends must have one row per row of x")
        d = _desc(x, K, w1.detach(), b1c, w2c.detach(), b2c, gc, bc, eps, gac, ia, gbc, ib, resc, out, out_res)
        blob = _prepared(d, K, (w1, b1c, w2c, b2c, gc, bc), x.device, owners=owners if owners is not None else (w1, b1, w2, b2, gamma, beta))
        d.prepared = blob.data_ptr()
        rows_io = 1 + (out is not None) + 2 * (out_res is not None)
        gathered = sum(min(R, t.shape[0]) for t in (gac, gbc) if t is not None)
        L.call("p4c_row_mlp_fwd", ctypes.byref(d), L.stream(x.device),
               alg_bytes=R * K * 2 + (rows_io - 1) * R * 128 + gathered * 128 + 4 * R * ((ga is not None) + (gb is not None)))
        ctx.save_for_backward(x, w1, w2c, b1c, b2c, gc, bc, gac, gbc)
        ctx.edges, ctx.eps, ctx.sinks, ctx.prepared_blob = edges, eps, sinks, blob
        ctx.flags = (b1 is not None, b2 is not None, gamma is not None, res is not None)
        ctx.pdtype = w1.dtype
        if out is None:
            ctx.mark_non_differentiable()
        return out, out_res

    @staticmethod
    def backward(ctx, dout, dout_res):
        x, w1, w2, b1, b2, gamma, beta, ga, gb = ctx.saved_tensors
        has_b1, has_b2, has_ln, has_res = ctx.flags
        edges = ctx.edges
        R, K = x.shape
        dy = None if dout is None else dout.contiguous()
        dyr = None if dout_res is None else dout_res.contiguous()
        if dy is None and dyr is None:
            return (None,) * 16
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty_like(x) if need_dx else None
        gathered = ga is not None or gb is not None
        dpre = torch.empty(R, 64, dtype=x.dtype, device=x.device) if gathered else None
        grads = None if ctx.sinks is not None else torch.empty(64 * K + 64 * 64 + 4 * 64, dtype=torch.float32, device=x.device)
        ws = torch.empty(max(L.lib().p4c_row_mlp_bwd_workspace_bytes(R, K) // 4, 1), dtype=torch.float32, device=x.device)
        ia = edges.src if (ga is not None and edges is not None) else None
        ib = edges.dst if (gb is not None and edges is not None) else None
        d = _desc(x, K, w1.detach(), b1, w2, b2, gamma, beta, ctx.eps, ga, ia, gb, ib, None, None, None, dy, dyr, dx, dpre)
        blob = ctx.prepared_blob   # the images the forward used (the parameters cannot have changed in between: autograd checks)
        d.prepared = blob.data_ptr()
        # res IS x (an edge update): the kernel stores dx + dy_res, the whole gradient of that tensor (no element-wise launch of autograd's)
        fold_res = ctx.res_is_x and has_res and need_dx and dyr is not None
        d.dx_plus_dy_res = int(fold_res)
        rows_io = 1 + (dy is not None) + (dyr is not None) + need_dx * K / 64 + gathered
        n_gath = sum(min(R, t.shape[0]) for t in (ga, gb) if t is not None)
        nbytes = R * K * 2 + (rows_io - 1) * R * 128 + n_gath * 128 + 4 * R * ((ga is not None) + (gb is not None))
        if ctx.sinks is not None:
            # parameter gradients go straight into the parameters' .grad buffers (+=): nothing is returned for them
            sw1, sb1, sw2, sb2, sg, sb = ctx.sinks
            p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
            gs = RowMlpGradSinks(dw1=p(sw1), ld_dw1=0 if sw1 is None else sw1.stride(0), db1=p(sb1), dw2=p(sw2), db2=p(sb2),
                                 dgamma=p(sg), dbeta=p(sb))
            from .ops_nodeproj import GradQueue

            GradQueue.begin(ws)    # the partials' reduction joins the batched ones at the end of this backward pass
            L.call("p4c_row_mlp_bwd_accumulate", ctypes.byref(d), ctypes.byref(gs), L.ptr(ws), L.stream(x.device), alg_bytes=nbytes)
            GradQueue.wrote(ctx.sinks)
            dw1 = dw2 = db1 = db2 = dgam = dbet = None
        else:
            L.call("p4c_row_mlp_bwd", ctypes.byref(d), L.ptr(grads), L.ptr(ws), L.stream(x.device), alg_bytes=nbytes)
            kr, o = w1.shape[1], w2.shape[0]
            dw1 = grads[: 64 * K].view(64, K)[:, :kr]
            dw2 = grads[64 * K: 64 * K + 4096].view(64, 64)[:o]
            base = 64 * K + 4096
            db1 = grads[base: base + 64] if has_b1 else None
            db2 = grads[base + 64: base + 64 + o] if has_b2 else None
            dgam = grads[base + 128: base + 192] if has_ln else None
            dbet = grads[base + 192: base + 256] if has_ln else None
        if edges is None:   # row-aligned addends: their gradient is the pre-activation gradient itself
            dga = dpre if (ga is not None and ctx.needs_input_grad[7]) else None
            dgb = dpre if (gb is not None and ctx.needs_input_grad[8]) else None
        else:
            want_a, want_b = ga is not None and ctx.needs_input_grad[7], gb is not None and ctx.needs_input_grad[8]
            if want_a and want_b:   # both adjoints of the gathers read the same dpre rows: one launch
                dga, dgb = _segment_sum_pair_raw(dpre, edges.by_src, edges.n_src, edges.by_dst, edges.n_dst)
            else:
                dga = _segment_sum_raw(dpre, *edges.by_src, edges.n_src) if want_a else None
                dgb = _segment_sum_raw(dpre, *edges.by_dst, edges.n_dst) if want_b else None
        dres = dyr if (has_res and not fold_res) else None
        return dx, dw1, db1, dw2, db2, dgam, dbet, dga, dgb, dres, None, None, None, None, None, None


def row_mlp(x: torch.Tensor, w1: torch.Tensor, b1, w2: torch.Tensor, b2, gamma=None, beta=None, eps: float = 1e-5,
            ga: Optional[torch.Tensor] = None, gb: Optional[torch.Tensor] = None, edges: Optional[EdgeSet] = None,
            res: Optional[torch.Tensor] = None, want_out: bool = True,
            grads_in_place: bool = False) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """Returns (y, y + res); x (R, K) bf16 with K <= 80 (padded to a multiple of 16 here), w1 (64, K), w2 (O <= 64, 64).
    ga / gb: (n_src, 64) / (n_dst, 64) rows added to the pre-activation through edges.src / edges.dst; with ``edges=None`` they
    are row-aligned addends (R, 64) -- e.g. the other half of a Linear over a concatenation of two 64-feature sources."""
    L.require_cuda(x)
    res_is_x = res is x and x.shape[1] == 64
    kp = (-x.shape[1]) % 16
    if kp:
        x = F.pad(x, (0, kp))
    sinks = grad_sinks(w1, b1, w2, b2, gamma, beta) if grads_in_place else None
    owners = (w1, b1, w2, b2, gamma, beta)   # the caller's parameter objects: what the weight-image cache checks for identity
    if sinks is not None:
        # autograd must not also accumulate what the kernel adds itself: five of the six parameters enter detached.  w1 stays
        # LIVE so that the node is recorded even when neither x nor the addends / residual need a gradient (first AR step of an
        # embedder: all inputs are data) -- its slot returns None in the backward, as _RowLinearSink does for its weight
        b1, w2, b2, gamma, beta = (None if t is None else t.detach() for t in (b1, w2, b2, gamma, beta))
    return _RowMLP.apply(x, w1, b1, w2, b2, gamma, beta, ga, gb, res, edges, eps, want_out, sinks, owners, res_is_x)


from .ops_rows import grad_view as _grad_view  # noqa: E402  (one definition, shared with ops_rows.row_linear)


def grad_sinks(w1, b1, w2, b2, gamma, beta):
    """Destinations for p4c_row_mlp_bwd_accumulate, or None when any parameter has no gradient buffer yet (e.g. before the first
    backward, or after zero_grad(set_to_none=True)) -- the caller then takes the ordinary autograd path."""
    if not torch.is_grad_enabled():
        return None
    views = [_grad_view(t) for t in (w1, b1, w2, b2, gamma, beta)]
    if any(v is False for v in views):
        return None
    if any(t is not None and not t.requires_grad for t in (w1, b1, w2, b2, gamma, beta)):
        return None
    if views[0].stride(1) != 1 or not views[2].is_contiguous():
        return None
    return tuple(views)
