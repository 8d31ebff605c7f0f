"""Autograd wrappers of the wide-channel GEMM / implicit-GEMM convolution kernels (csrc/gemm.hip; include/py4cast_hip.h:
p4c_gemm_prep_weight, p4c_gemm_nt, p4c_gemm_tn, p4c_bnorm_finalize).

They carry what mfai's UNETR++ / SwinUNETR (py4cast/models.py:10-20; config/CLI/model/unetrpp.yaml:19-35, swinunetr.yaml:19-30) express as
``nn.Linear`` / ``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.GELU`` calls, on features-last bf16 activations with fp32 master parameters:

* ``linear(x, w, b, res)``            y = x W^T + b (+ res)                       -- forward, data gradient, weight + bias gradient native
* ``mlp(x, w1, b1, w2, b2, res)``     y = gelu(x W1^T + b1) W2^T + b2 (+ res)     -- ONE autograd node: GELU in the first product's
  epilogue, GELU' in the epilogue of the second product's data gradient (no elementwise launch either way)
* ``conv3x3(x, w, res, stats)``       3x3 "same" convolution of an NHWC map, no im2col buffer; optionally the per-channel sums of the
  output for the batch norm that follows
* ``batch_norm_act(y, stats, ...)``   BatchNorm2d (training statistics from the producer's sums, running statistics updated) +
  LeakyReLU (+ residual) as one node on the streaming kernels of csrc/inorm.hip (a batch norm is an instance norm of the batch seen
  as one sample)

No CPU fallback: every entry point raises on CPU tensors."""

from typing import Optional, Tuple

import torch

from . import _lib as L

ACT_NONE, ACT_GELU_FWD, ACT_GELU_BWD = 0, 1, 2

_WIMG = {}   # eager mode: (address, shape, strides, taps) -> (parameter version, (fwd image, dgrad image), owners)


# Round 6: the images of ALL weights a model used before are prepared at the first miss of a step (eager: after the optimizer moved the
# parameters; a HIP-graph capture: at the first use inside it), 24 per launch (p4c_gemm_prep_weight_batch) -- one launch per weight before:
# 154 per UNETR++ step, 54 per SwinUNETR step.  The log holds what was asked for (weak references to the parameters, in first-use order);
# an entry is prepared again only when its cache entry is stale by the same test a single lookup applies.
_PREP_LOG = {}        # cache key -> (weakref(w), taps, None | (weakref(scale parameter), first, length), None | weakref(b) | False)
BATCHED_PREP = True


def _whole_parameter(t: torch.Tensor):
    """the parameter behind t when t IS that parameter (or a rollout's stand-in of it), else None (slices are prepared one by one)"""
    base = L._base(t)
    if base.data_ptr() == t.data_ptr() and base.shape == t.shape and base.stride() == t.stride() and base.dtype == t.dtype:
        return base
    return None


def _plain_key(w, taps):
    return ("wimg", w.data_ptr(), tuple(w.shape), tuple(w.stride()), taps)


def _scaled_key(w, b, g):
    return ("wimg_scaled", w.data_ptr(), tuple(w.shape), tuple(w.stride()), g.data_ptr(), None if b is None else b.data_ptr())


def _fresh(cache, key, ver, owners) -> bool:
    hit = cache.get(key)
    return hit is not None and hit[0] == ver and L.owners_alive(hit[2], owners)


def _prepare_logged(cache, device) -> None:
    """prepare the images of every logged weight whose entry in ``cache`` is stale: one launch per 24 jobs"""
    import ctypes

    jobs, dead = [], []
    for key, (rw, taps, rg, rb) in _PREP_LOG.items():
        w = rw()
        scaled = rg is not None
        g = b = None
        gone = w is None
        if scaled and not gone:
            gbase = rg[0]()                       # (the scale is a run of a 1-D parameter: base, first element, length)
            gone = gbase is None
            if not gone:
                with torch.no_grad():
                    g = gbase[rg[1]: rg[1] + rg[2]]          # (a view OF THE PARAMETER: the cache's owner test sees what a caller's slice shows)
            if rb is not False and not gone:
                b = rb()
                gone = b is None
        if gone:
            dead.append(key)
            continue
        if w.device != device:
            continue
        if (key != (_scaled_key(w, b, g) if scaled else _plain_key(w, taps))) or w.dtype != torch.float32 or not w.is_contiguous():
            dead.append(key)          # the parameter moved (another storage / layout): it is logged again under its new key when used
            continue
        if scaled:
            ver = (L.PARAM_EPOCH[0], w._version, g._version, None if b is None else b._version)
            owners = (w, g) if b is None else (w, g, b)
        else:
            ver, owners = (L.PARAM_EPOCH[0], w._version), (w,)
        if _fresh(cache, key, ver, owners):
            continue
        CO, CI = w.shape[0], w.shape[1]
        fwd = torch.empty(CO, taps * CI, dtype=torch.bfloat16, device=device)
        dgr = torch.empty(CI, taps * CO, dtype=torch.bfloat16, device=device)
        beff = torch.empty(CO, dtype=torch.float32, device=device) if (scaled and b is not None) else None
        jobs.append((key, ver, owners, w.detach(), None if g is None else g.detach(), None if b is None else b.detach(), beff, CO, CI, taps, fwd, dgr,
                     scaled))
    for key in dead:
        del _PREP_LOG[key]
    if not jobs:
        return
    n = len(jobs)
    PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
    ptr = lambda t: None if t is None else t.data_ptr()     # noqa: E731
    L.call("p4c_gemm_prep_weight_batch", n, PA(*[ptr(j[3]) for j in jobs]), PA(*[ptr(j[4]) for j in jobs]), PA(*[ptr(j[5]) for j in jobs]),
           PA(*[ptr(j[6]) for j in jobs]), IA(*[j[7] for j in jobs]), IA(*[j[8] for j in jobs]), IA(*[j[9] for j in jobs]),
           PA(*[ptr(j[10]) for j in jobs]), PA(*[ptr(j[11]) for j in jobs]), L.stream(device))
    for key, ver, owners, _w, _g, _b, beff, _co, _ci, _taps, fwd, dgr, scaled in jobs:
        refs = L.owner_refs(owners)
        assert L.owners_alive(refs, owners)
        cache[key] = (ver, (fwd, dgr, beff) if scaled else (fwd, dgr), refs)


def weight_images(w: torch.Tensor, taps: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """bf16 operand images of an fp32 master weight (CO, CI[, 3, 3]): ([CO][taps][CI], [CI][taps'][CO]); once per parameter version
    in eager mode and once per HIP-graph capture (the AR steps of a rollout share them)."""
    L.require_cuda(w)
    if w.dtype != torch.float32:
        raise L.P4CError(f"weight_images: fp32 master weights expected, got {w.dtype}")
    wd = w.detach()
    if not wd.is_contiguous():
        wd = wd.contiguous()
    CO, CI = wd.shape[0], wd.shape[1]
    key = _plain_key(w, taps)
    ver = (L.PARAM_EPOCH[0], w._version)
    cache = _WIMG if not torch.cuda.is_current_stream_capturing() else L.capture_cache()
    if cache is not None:
        if _fresh(cache, key, ver, (w,)):
            return cache[key][1]
        if BATCHED_PREP and w.is_contiguous():
            import weakref

            base = _whole_parameter(w)
            if base is not None:
                if key not in _PREP_LOG:
                    _PREP_LOG[key] = (weakref.ref(base), taps, None, None)
                _prepare_logged(cache, w.device)
                if _fresh(cache, key, ver, (w,)):
                    return cache[key][1]
    fwd = torch.empty(CO, taps * CI, dtype=torch.bfloat16, device=w.device)
    dgr = torch.empty(CI, taps * CO, dtype=torch.bfloat16, device=w.device)
    L.call("p4c_gemm_prep_weight", L.ptr(wd), CO, CI, taps, L.ptr(fwd), L.ptr(dgr), L.stream(w.device))
    if cache is not None:
        cache[key] = (ver, (fwd, dgr), L.owner_refs((w,)))
    return fwd, dgr


def scaled_images(w: torch.Tensor, b: Optional[torch.Tensor], g: torch.Tensor):
    """(forward image, data-gradient image, g * b or None) of the layer ``g (.) (x W^T + b)`` -- a per-output-channel scale (UNETR++'s
    layer scale) folded into the weight images by the preparation kernel; cached like weight_images on the versions of w, b and g."""
    L.require_cuda(w)
    if w.dtype != torch.float32 or g.dtype != torch.float32 or (b is not None and b.dtype != torch.float32):
        raise L.P4CError("scaled_images: fp32 master parameters expected")
    wd, gd = w.detach().contiguous(), g.detach().contiguous()
    bd = None if b is None else b.detach().contiguous()
    CO, CI = wd.shape
    key = _scaled_key(w, b, g)
    ver = (L.PARAM_EPOCH[0], w._version, g._version, None if b is None else b._version)
    owners = (w, g) if b is None else (w, g, b)
    cache = _WIMG if not torch.cuda.is_current_stream_capturing() else L.capture_cache()
    if cache is not None:
        if _fresh(cache, key, ver, owners):
            return cache[key][1]
        if BATCHED_PREP and w.is_contiguous() and g.is_contiguous() and (b is None or b.is_contiguous()):
            import weakref

            # (g is a run of the 1-D layer-scale parameter, w and b whole parameters -- or a rollout's stand-ins of them)
            bw, gb = _whole_parameter(w), L._base(g)
            bb = None if b is None else _whole_parameter(b)
            if bw is not None and g.dim() == 1 and gb.dim() == 1 and gb.is_contiguous() and (b is None or bb is not None):
                if key not in _PREP_LOG:
                    _PREP_LOG[key] = (weakref.ref(bw), 1, (weakref.ref(gb), g.storage_offset() - gb.storage_offset(), g.numel()),
                                      False if b is None else weakref.ref(bb))
                _prepare_logged(cache, w.device)
                if _fresh(cache, key, ver, owners):
                    return cache[key][1]
    fwd = torch.empty(CO, CI, dtype=torch.bfloat16, device=w.device)
    dgr = torch.empty(CI, CO, dtype=torch.bfloat16, device=w.device)
    beff = None if b is None else torch.empty(CO, dtype=torch.float32, device=w.device)
    L.call("p4c_gemm_prep_weight_scaled", L.ptr(wd), L.ptr(gd), L.ptr(bd), L.ptr(beff), CO, CI, 1, L.ptr(fwd), L.ptr(dgr), L.stream(w.device))
    if cache is not None:
        cache[key] = (ver, (fwd, dgr, beff), L.owner_refs(owners))
    return fwd, dgr, beff


def _rows(t: torch.Tensor, C: int) -> torch.Tensor:
    """(..., C) -> (R, C) rows with unit stride inside a row, a row stride that is a multiple of 8 and a 16-byte aligned base"""
    t2 = t.reshape(-1, C)
    if t2.stride(1) != 1 or t2.stride(0) % 8 or t2.data_ptr() % 16:
        t2 = t2.contiguous()
    return t2


def _f32(p: Optional[torch.Tensor]):
    if p is None:
        return None
    p = p.detach()
    return p.contiguous() if p.dtype == torch.float32 else p.float().contiguous()


def gemm_nt(A: torch.Tensor, img: torch.Tensor, N: int, K: int, conv=None, bias=None, res=None, act=ACT_NONE, aux_in=None,
            want_stats=False, out=None):
    """C = epilogue(A x img^T).  A: (M, >= K) bf16 rows, or with conv = (H, W, Cin) the NHWC map as (M = batch H W, Cin-or-wider)
    rows.  Returns (C, aux_out or None, stats or None); `out`: a preallocated (M, >= N) row view to write into."""
    lib = L.lib()
    M = A.shape[0]
    H, W, Cin, taps = (conv[0], conv[1], conv[2], 9) if conv is not None else (0, 0, 0, 1)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=A.device) if out is None else out
    aux_out = torch.empty(M, N, dtype=torch.bfloat16, device=A.device) if act == ACT_GELU_FWD else None
    aux = aux_out if act == ACT_GELU_FWD else aux_in
    stats = None
    if want_stats:
        stats = torch.empty(lib.p4c_gemm_nt_stat_blocks(M, N, K), 2, N, dtype=torch.float32, device=A.device)
    nbytes = lib.p4c_gemm_nt_workspace_bytes(M, N, K)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=A.device) if nbytes else None
    L.call("p4c_gemm_nt", L.ptr(A), A.stride(0), L.ptr(img), M, N, K, H, W, Cin, taps, L.ptr(bias), L.ptr(res),
           0 if res is None else res.stride(0), act, L.ptr(aux_in), L.ptr(aux_out), 0 if aux is None else aux.stride(0), L.ptr(C), C.stride(0),
           L.ptr(stats), L.ptr(ws), L.stream(A.device), alg_bytes=2 * (M * (K if conv is None else Cin) + N * K + M * N), alg_flops=2 * M * N * K)
    return C, aux_out, stats


def gemm_tn(dy: torch.Tensor, x: torch.Tensor, Mo: int, Cin: int, conv=None, want_bias=False, sink=None):
    """dW (Mo, Cin[, 3, 3]) fp32 and db (Mo) or None from dy (R, >= Mo) and x (R, >= Cin) bf16 rows (conv = (H, W): x is the NHWC map).
    sink = (gw, gb): ADD both into these fp32 buffers (views of the parameters' .grad) instead, and return (None, None)."""
    lib = L.lib()
    R = dy.shape[0]
    taps, H, W = (9, conv[0], conv[1]) if conv is not None else (1, 0, 0)
    if sink is not None:
        dw, db = sink
    else:
        dw = torch.empty((Mo, Cin, 3, 3) if conv is not None else (Mo, Cin), dtype=torch.float32, device=dy.device)
        db = torch.empty(Mo, dtype=torch.float32, device=dy.device) if want_bias else None
    ws = torch.empty(max(lib.p4c_gemm_tn_workspace_bytes(R, Mo, taps * Cin) // 4, 1), dtype=torch.float32, device=dy.device)
    if sink is not None:
        # round 6: inside a backward pass the split-K slabs of the accumulating calls are summed 32 calls per launch when the pass ends
        # (csrc/gemm.hip: gemm_tn_reduce_batch_kernel; ops_nodeproj.GradQueue keeps the slabs alive and flushes) -- 804 reduce launches per
        # UNETR++ step before.  Not when a gradient exchange listens for the sums as they become final (FlatDDP(overlap=True), eager
        # steps of N > 1 ranks): there every call reduces at once and reports it.
        from .ops_nodeproj import GradQueue

        if GradQueue.active() or not L.GRAD_SINK_LISTENERS:
            GradQueue.begin(ws)
    L.call("p4c_gemm_tn", L.ptr(dy), dy.stride(0), L.ptr(x), x.stride(0), R, Mo, H, W, Cin, taps, L.ptr(dw), L.ptr(db), int(sink is not None),
           L.ptr(ws), L.stream(dy.device), alg_bytes=2 * R * (Mo + Cin) + 4 * Mo * Cin * taps, alg_flops=2 * R * Mo * Cin * taps)
    if sink is not None:
        GradQueue.wrote([dw, db])    # (FlatDDP(overlap=True) counts these: now, or at the flush when the sums were queued)
        return None, None
    return dw, db


# Weight / bias gradients are ADDED straight into the parameters' .grad buffers by the reduction kernel (p4c_gemm_tn, accumulate = 1)
# instead of being returned for autograd's AccumulateGrad -- one small `+=` launch per parameter and AR step otherwise (and per
# stand-in of trainer.RolloutParamProxies).  Only when the parameter (and its bias) already HAS a contiguous fp32 gradient buffer
# (FlatDDP / the trainer allocate them; otherwise the ordinary path runs); set to False for code that needs torch.autograd.grad()
# with respect to these parameters.  Same convention as graphlam.GRADS_IN_PLACE.
GRADS_IN_PLACE = True


def _sink(w, b):
    """(gw, gb) views of the .grad buffers of a weight and its bias (gb None without a bias), or None when the gradients must go
    through autograd"""
    # (called inside Function.forward, where grad mode is off: requires_grad of the inputs says whether a backward will come)
    if not (GRADS_IN_PLACE and w.requires_grad and (b is None or b.requires_grad)):
        return None
    from .ops_rows import grad_view

    gw = grad_view(w)
    gb = None if b is None else grad_view(b)
    if gw is None or gw is False or gb is False or not gw.is_contiguous() or (gb is not None and not gb.is_contiguous()):
        return None
    return gw, gb


def supported(x: torch.Tensor, w: torch.Tensor) -> bool:
    """bf16 activations on the GPU, fp32 master weight, channel counts on the kernels' 8-element granularity"""
    return (x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.float32 and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, res):
        O, K = w.shape
        x2 = _rows(x.detach(), K)
        fwd, dgr = weight_images(w, 1)
        r2 = None if res is None else _rows(res.detach(), O)
        y, _, _ = gemm_nt(x2, fwd, O, K, bias=_f32(b), res=r2)
        ctx.save_for_backward(x2, dgr)
        ctx.xshape, ctx.has_bias, ctx.has_res = x.shape, b is not None, res is not None
        ctx.wdtype, ctx.bdtype, ctx.OK = w.dtype, (None if b is None else b.dtype), (O, K)
        ctx.sink = _sink(w, b)
        return y.view(*x.shape[:-1], O)

    @staticmethod
    def backward(ctx, dy):
        x2, dgr = ctx.saved_tensors
        O, K = ctx.OK
        dy2 = _rows(dy, O)
        dx = gemm_nt(dy2, dgr, K, O)[0].view(ctx.xshape) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = gemm_tn(dy2, x2, O, K, want_bias=ctx.has_bias, sink=ctx.sink)
            dw = None if dw is None else dw.to(ctx.wdtype)
            db = None if db is None else db.to(ctx.bdtype)
        return dx, dw, db, (dy if ctx.has_res else None)


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``F.linear(x, w, b) (+ res)`` for bf16 activations of any rank with fp32 master parameters (in / out features multiples of 8)"""
    L.require_cuda(x)
    if not supported(x, w):
        raise L.P4CError(f"ops_gemm.linear: unsupported operands ({x.dtype}, weight {tuple(w.shape)} {w.dtype})")
    return _Linear.apply(x, w, b, res)


class _CatLinearRes(torch.autograd.Function):
    """res + gamma (.) cat(xa Wa^T + ba, xb Wb^T + bb) along the features (gamma optional): two GEMMs whose epilogues write the two
    column halves of ONE output tensor (row stride = the full width) and add the matching halves of the residual.  gamma rides in the
    weight images and the bias (scaled_images: prepared once per step, not once per AR step); backward takes the RAW weight / bias
    gradients of the unscaled dy and one small launch per half turns them into dW, db and dgamma (p4c_gemm_scale_fold_bwd) -- as
    separate autograd nodes the four products gamma * W, gamma * b cost 4 launches forward and ~16 backward per block and AR step."""

    @staticmethod
    def forward(ctx, xa, wa, ba, xb, wb, bb, res, gamma):
        Oa, Ka = wa.shape
        Ob, Kb = wb.shape
        xa2, xb2 = _rows(xa.detach(), Ka), _rows(xb.detach(), Kb)
        r2 = _rows(res.detach(), Oa + Ob)
        if gamma is None:
            fa, da = weight_images(wa, 1)
            fb, db_ = weight_images(wb, 1)
            bea, beb = _f32(ba), _f32(bb)
        else:
            fa, da, bea = scaled_images(wa, ba, gamma[:Oa])
            fb, db_, beb = scaled_images(wb, bb, gamma[Oa:])
        y = torch.empty(xa2.shape[0], Oa + Ob, dtype=torch.bfloat16, device=xa.device)
        gemm_nt(xa2, fa, Oa, Ka, bias=bea, res=r2[:, :Oa], out=y[:, :Oa])
        gemm_nt(xb2, fb, Ob, Kb, bias=beb, res=r2[:, Oa:], out=y[:, Oa:])
        ctx.meta = (xa.shape, xb.shape, Oa, Ka, Ob, Kb, wa.dtype, ba.dtype, wb.dtype, bb.dtype)
        ctx.sinks = (_sink(wa, ba), _sink(wb, bb))
        ctx.scaled = gamma is not None
        if ctx.scaled:
            from .ops_rows import grad_view

            gg = grad_view(gamma) if (GRADS_IN_PLACE and gamma.requires_grad) else False
            ctx.gsink = gg if (gg is not None and gg is not False and gg.is_contiguous()) else None
            ctx.save_for_backward(xa2, xb2, da, db_, wa.detach(), ba.detach(), wb.detach(), bb.detach(), gamma.detach())
        else:
            ctx.save_for_backward(xa2, xb2, da, db_)
        return y.view(*res.shape)

    @staticmethod
    def backward(ctx, dy):
        sa, sb, Oa, Ka, Ob, Kb, wadt, badt, wbdt, bbdt = ctx.meta
        dy2 = _rows(dy, Oa + Ob)
        dya, dyb = dy2[:, :Oa], dy2[:, Oa:]
        cast = lambda t, dt: None if t is None else t.to(dt)      # noqa: E731
        if not ctx.scaled:
            xa2, xb2, da, db_ = ctx.saved_tensors
            dxa = gemm_nt(dya, da, Ka, Oa)[0].view(sa)
            dxb = gemm_nt(dyb, db_, Kb, Ob)[0].view(sb)
            dwa, dba = gemm_tn(dya, xa2, Oa, Ka, want_bias=True, sink=ctx.sinks[0])
            dwb, dbb = gemm_tn(dyb, xb2, Ob, Kb, want_bias=True, sink=ctx.sinks[1])
            return dxa, cast(dwa, wadt), cast(dba, badt), dxb, cast(dwb, wbdt), cast(dbb, bbdt), dy, None
        xa2, xb2, da, db_, wa, ba, wb, bb, gamma = ctx.saved_tensors
        dxa = gemm_nt(dya, da, Ka, Oa)[0].view(sa)          # (the images carry gamma: dx = (gamma (.) dy) W)
        dxb = gemm_nt(dyb, db_, Kb, Ob)[0].view(sb)
        in_place = ctx.sinks[0] is not None and ctx.sinks[1] is not None and ctx.gsink is not None
        dgamma = ctx.gsink if in_place else torch.empty(Oa + Ob, dtype=torch.float32, device=dy.device)
        outs = []
        for dyh, xh, w, b, O, K, sink, g, dg in ((dya, xa2, wa, ba, Oa, Ka, ctx.sinks[0], gamma[:Oa], dgamma[:Oa]),
                                                  (dyb, xb2, wb, bb, Ob, Kb, ctx.sinks[1], gamma[Oa:], dgamma[Oa:])):
            dw_raw, db_raw = gemm_tn(dyh, xh, O, K, want_bias=True)
            dw, db = sink if in_place else (torch.empty_like(dw_raw), torch.empty_like(db_raw))
            L.call("p4c_gemm_scale_fold_bwd", L.ptr(dw_raw), L.ptr(db_raw), L.ptr(_f32(w)), L.ptr(_f32(b)), L.ptr(_f32(g)), O, K, L.ptr(dw), L.ptr(db),
                   L.ptr(dg), int(in_place), L.stream(dy.device), alg_bytes=4 * O * K * (3 + in_place))
            if in_place:
                L.grad_written(dw, db, dg)
            outs.append((None, None) if in_place else (dw, db))
        (dwa, dba), (dwb, dbb) = outs
        return (dxa, cast(dwa, wadt), cast(dba, badt), dxb, cast(dwb, wbdt), cast(dbb, bbdt), dy,
                None if in_place else dgamma.to(gamma.dtype))


def cat_linear_res(xa, wa, ba, xb, wb, bb, res, gamma=None) -> torch.Tensor:
    """``res + gamma * torch.cat([F.linear(xa, wa, ba), F.linear(xb, wb, bb)], -1)`` (gamma: a vector over the features or None) as one
    autograd node (bf16 rows, fp32 parameters)"""
    L.require_cuda(xa)
    if not (supported(xa, wa) and supported(xb, wb)) or (wa.shape[0] % 8) or (wb.shape[0] % 8):
        raise L.P4CError("ops_gemm.cat_linear_res: unsupported operands")
    if gamma is not None and (ba is None or bb is None or gamma.dtype != torch.float32 or gamma.numel() != wa.shape[0] + wb.shape[0]):
        raise L.P4CError("ops_gemm.cat_linear_res: gamma needs fp32 parameters with biases and one entry per output feature")
    return _CatLinearRes.apply(xa, wa, ba, xb, wb, bb, res, gamma)


class _MLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, res):
        Hd, K = w1.shape
        O = w2.shape[0]
        x2 = _rows(x.detach(), K)
        f1, d1 = weight_images(w1, 1)
        f2, d2 = weight_images(w2, 1)
        g, h, _ = gemm_nt(x2, f1, Hd, K, bias=_f32(b1), act=ACT_GELU_FWD)
        r2 = None if res is None else _rows(res.detach(), O)
        y, _, _ = gemm_nt(g, f2, O, Hd, bias=_f32(b2), res=r2)
        ctx.save_for_backward(x2, g, h, d1, d2)
        ctx.xshape, ctx.dims, ctx.has_res = x.shape, (K, Hd, O), res is not None
        ctx.dt = (w1.dtype, None if b1 is None else b1.dtype, w2.dtype, None if b2 is None else b2.dtype)
        ctx.sinks = (_sink(w1, b1), _sink(w2, b2))
        return y.view(*x.shape[:-1], O)

    @staticmethod
    def backward(ctx, dy):
        x2, g, h, d1, d2 = ctx.saved_tensors
        K, Hd, O = ctx.dims
        dy2 = _rows(dy, O)
        dh = gemm_nt(dy2, d2, Hd, O, act=ACT_GELU_BWD, aux_in=h)[0]            # (dy W2) * gelu'(h)
        dw2, db2 = gemm_tn(dy2, g, O, Hd, want_bias=ctx.dt[3] is not None, sink=ctx.sinks[1])
        dx = gemm_nt(dh, d1, K, Hd)[0].view(ctx.xshape) if ctx.needs_input_grad[0] else None
        dw1, db1 = gemm_tn(dh, x2, Hd, K, want_bias=ctx.dt[1] is not None, sink=ctx.sinks[0])
        cast = lambda t, dt: None if t is None else t.to(dt)      # noqa: E731
        return (dx, cast(dw1, ctx.dt[0]), cast(db1, ctx.dt[1]), cast(dw2, ctx.dt[2]), cast(db2, ctx.dt[3]), (dy if ctx.has_res else None))


def mlp(x, w1, b1, w2, b2, res=None) -> torch.Tensor:
    """``F.linear(F.gelu(F.linear(x, w1, b1)), w2, b2) (+ res)`` as one autograd node"""
    L.require_cuda(x)
    if not (supported(x, w1) and supported(x, w2)):
        raise L.P4CError("ops_gemm.mlp: unsupported operands")
    return _MLP.apply(x, w1, b1, w2, b2, res)


class _Conv(torch.autograd.Function):
    """y (B,H,W,Co) = conv(x (B,H,W,>=Ci), w (Co,Ci,k,k)) (+ bias) (+ res), k = 3 ("same", zero padding) or 1; second output: the column
    sums of y for a batch norm (not differentiable: the norm's backward accounts for them analytically)."""

    @staticmethod
    def forward(ctx, x, w, b, res, want_stats, passthrough=False):
        Co, Ci, k = w.shape[0], w.shape[1], w.shape[2]
        B, H, W_, Cx = x.shape
        taps = k * k
        xm = _rows(x.detach(), Cx)
        fwd, dgr = weight_images(w, taps)
        r2 = None if res is None else _rows(res.detach(), Co)
        if taps == 9:
            y, _, stats = gemm_nt(xm, fwd, Co, 9 * Ci, conv=(H, W_, Ci), bias=_f32(b), res=r2, want_stats=want_stats)
        else:
            y, _, stats = gemm_nt(xm, fwd, Co, Ci, bias=_f32(b), res=r2, want_stats=want_stats)
        ctx.save_for_backward(xm, dgr)
        ctx.geom, ctx.has_bias, ctx.has_res = (B, H, W_, Cx, Co, Ci, taps), b is not None, res is not None
        ctx.wdtype, ctx.bdtype, ctx.wshape = w.dtype, (None if b is None else b.dtype), w.shape
        ctx.sink = _sink(w, b)
        ctx.set_materialize_grads(False)     # (no zero-filled "gradient" of the statistics output: a fill launch per convolution)
        y = y.view(B, H, W_, Co)
        if want_stats:
            ctx.mark_non_differentiable(stats)
        # passthrough: x itself is one more output -- the tensor's OTHER consumers (a residual connection) take it from here, so their
        # gradient arrives at this node and is added in the data gradient's epilogue instead of by a launch of autograd's
        return y, (stats if want_stats else None), (x if passthrough else None)

    @staticmethod
    def backward(ctx, dy, _dstats, dpass):
        if dy is None:
            return dpass, None, None, None, None, None
        xm, dgr = ctx.saved_tensors
        B, H, W_, Cx, Co, Ci, taps = ctx.geom
        dy2 = _rows(dy, Co)
        dx = None
        if ctx.needs_input_grad[0]:
            fold = dpass is not None and Cx == Ci and dpass.dtype == dy.dtype
            r2 = _rows(dpass, Cx) if fold else None
            if taps == 9:
                dx = gemm_nt(dy2, dgr, Ci, 9 * Co, conv=(H, W_, Co), res=r2)[0]
            else:
                dx = gemm_nt(dy2, dgr, Ci, Co, res=r2)[0]
            if Cx > Ci:      # the map was wider than the weight's input channels (zero-padded rows): no gradient there
                dx = torch.nn.functional.pad(dx, (0, Cx - Ci))
            dx = dx.view(B, H, W_, Cx)
            if dpass is not None and not fold:
                dx = dx + dpass
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = gemm_tn(dy2, xm, Co, Ci, conv=(H, W_) if taps == 9 else None, want_bias=ctx.has_bias, sink=ctx.sink)
            dw = None if dw is None else dw.view(ctx.wshape).to(ctx.wdtype)
            db = None if db is None else db.to(ctx.bdtype)
        return dx, dw, db, (dy if ctx.has_res else None), None, None


def conv_supported(x: torch.Tensor, w: torch.Tensor) -> bool:
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and w.dtype == torch.float32 and w.dim() == 4
            and w.shape[2] == w.shape[3] and w.shape[2] in (1, 3) and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0
            and x.shape[-1] >= w.shape[1] and x.shape[-1] % 8 == 0)


def conv2d_nhwc(x, w, b=None, res=None, want_stats=False, passthrough=False):
    """3x3 "same" / 1x1 convolution of a features-last map; returns y, or (y, stats) with want_stats.  ``passthrough``: x is appended to
    the results -- hand THAT tensor to the other consumers of x (a residual connection): their gradient is then added inside the data
    gradient's epilogue."""
    L.require_cuda(x)
    if not conv_supported(x, w):
        raise L.P4CError(f"ops_gemm.conv2d_nhwc: unsupported operands (x {tuple(x.shape)} {x.dtype}, w {tuple(w.shape)} {w.dtype})")
    y, stats, xp = _Conv.apply(x, w, b, res, bool(want_stats), bool(passthrough))
    out = (y, stats) if want_stats else (y,)
    if passthrough:
        out = out + (xp,)
    return out if len(out) > 1 else y


class _BatchNormAct(torch.autograd.Function):
    """out = lrelu(batch_norm(y) (+ res), slope) for y (B,H,W,C) features-last: statistics over (B,H,W) per channel from the producer's
    partial sums (training) or the running statistics (eval); the streaming kernels of csrc/inorm.hip with the batch as ONE sample."""

    @staticmethod
    def forward(ctx, y, stats, gamma, beta, res, running_mean, running_var, training, momentum, eps, slope, batches_tracked=None,
                res_passthrough=False, mul=None, mul_factor=1.0):
        yc = y.contiguous()
        C = yc.shape[-1]
        N = yc.numel() // C
        dev = yc.device
        st = torch.empty(4, C, dtype=torch.float32, device=dev)
        g32, b32 = _f32(gamma), _f32(beta)
        if training:
            if stats is None:   # no producer sums: one reduction pass (p4c_inorm_reduce)
                nb = L.lib().p4c_inorm_blocks(N, C)
                stats = torch.empty(nb, 2, C, dtype=torch.float32, device=dev)
                L.call("p4c_inorm_reduce", L.ptr(yc), None, None, None, None, 1.0, L.ptr(stats), L.dtype_code(yc.dtype), 1, N, C, L.stream(dev))
            L.call("p4c_bnorm_finalize", L.ptr(stats), stats.shape[0], float(N), C, L.ptr(g32), L.ptr(b32), float(eps), float(momentum),
                   L.ptr(running_mean), L.ptr(running_var), L.ptr(st[0]), L.ptr(st[1]), L.ptr(st[2]), L.ptr(st[3]), L.ptr(batches_tracked), L.stream(dev))
        else:
            st[0] = running_mean
            st[1] = torch.rsqrt(running_var.float() + eps)
            st[2] = st[1] * (1.0 if g32 is None else g32)
            st[3] = (0.0 if b32 is None else b32) - st[0] * st[2]
        out = torch.empty_like(yc)
        rc = None if res is None else res.contiguous()
        # mul (groups, C) fp32 >= 0, mul_factor: a multiplier per (row group, channel) behind the activation -- a channel dropout's draw
        # per (sample, channel) and 1 / (1 - p) -- applied by THIS pass and its backward passes (csrc/inorm.hip: PostMul)
        ctx.mul_rows = 0
        if mul is not None:
            if mul.dtype != torch.float32 or mul.dim() != 2 or mul.shape[1] != C or not mul.is_contiguous() or N % mul.shape[0]:
                raise L.P4CError("ops_gemm.batch_norm_act: the multiplier is a dense fp32 (groups, C) table whose group count divides the rows")
            ctx.mul_rows, ctx.mul_factor = N // mul.shape[0], float(mul_factor)
            L.call("p4c_inorm_apply_mul", L.ptr(yc), L.ptr(rc), None, None, L.ptr(st[2]), L.ptr(st[3]), None, None, None, None, float(slope),
                   L.ptr(out), None, L.dtype_code(yc.dtype), 1, N, C, L.ptr(mul), ctx.mul_rows, ctx.mul_factor, L.stream(dev),
                   alg_bytes=yc.numel() * yc.element_size() * (2 + (rc is not None)))
        else:
            L.call("p4c_inorm_apply", L.ptr(yc), L.ptr(rc), None, None, L.ptr(st[2]), L.ptr(st[3]), None, None, None, None, float(slope),
                   L.ptr(out), None, L.dtype_code(yc.dtype), 1, N, C, L.stream(dev), alg_bytes=yc.numel() * yc.element_size() * (2 + (rc is not None)))
        ctx.save_for_backward(yc, out, st, *(() if mul is None else (mul,)))
        ctx.slope, ctx.has_res, ctx.training = float(slope), res is not None, bool(training)
        ctx.gdtype = None if gamma is None else gamma.dtype
        ctx.set_materialize_grads(False)
        # res_passthrough: the residual operand is returned as a second output for ITS other consumers (see _Conv): their gradient
        # arrives here and is added to the residual's gradient inside the backward apply launch
        return out, (res if (res_passthrough and res is not None) else None)

    @staticmethod
    def backward(ctx, dout, dpass):
        if dout is None:
            return (None, None, None, None, dpass) + (None,) * 10
        yc, out, st, *rest = ctx.saved_tensors
        mul = rest[0] if rest else None
        C = yc.shape[-1]
        N = yc.numel() // C
        dev = yc.device
        dout = dout.contiguous()
        from . import ops_inorm as ON

        nb = L.lib().p4c_inorm_blocks(N, C)
        part = torch.empty(1, nb, 2, C, dtype=torch.float32, device=dev)
        co = torch.empty(2, C, dtype=torch.float32, device=dev)
        dgb = torch.empty(2, C, dtype=torch.float32, device=dev)
        if mul is not None:
            L.call("p4c_inorm_reduce_mul", L.ptr(yc), L.ptr(dout), L.ptr(out), L.ptr(st[0]), L.ptr(st[1]), ctx.slope, L.ptr(part), L.dtype_code(yc.dtype),
                   1, N, C, L.ptr(mul), ctx.mul_rows, ctx.mul_factor, L.stream(dev), alg_bytes=3 * yc.numel() * yc.element_size())
            L.call("p4c_inorm_finalize_bwd", L.ptr(part), nb, 1, N, C, 0, None, None, L.ptr(co[0]), L.ptr(co[1]), L.ptr(dgb[0]), L.ptr(dgb[1]),
                   L.stream(dev))
        elif ON.FUSED_FINALIZE:    # the coefficients and dgamma / dbeta from the reduce launch's last workgroup (csrc/inorm.hip: InFin)
            L.call("p4c_inorm_reduce_finalize_bwd", L.ptr(yc), L.ptr(dout), L.ptr(out), L.ptr(st[0]), L.ptr(st[1]), ctx.slope, L.ptr(part),
                   ON.next_ticket(dev), L.ptr(co[0]), L.ptr(co[1]), L.ptr(dgb[0]), L.ptr(dgb[1]), L.dtype_code(yc.dtype), 1, N, C, L.stream(dev),
                   alg_bytes=3 * yc.numel() * yc.element_size())
        else:
            L.call("p4c_inorm_reduce", L.ptr(yc), L.ptr(dout), L.ptr(out), L.ptr(st[0]), L.ptr(st[1]), ctx.slope, L.ptr(part), L.dtype_code(yc.dtype),
                   1, N, C, L.stream(dev), alg_bytes=3 * yc.numel() * yc.element_size())
            L.call("p4c_inorm_finalize_bwd", L.ptr(part), nb, 1, N, C, 0, None, None, L.ptr(co[0]), L.ptr(co[1]), L.ptr(dgb[0]), L.ptr(dgb[1]),
                   L.stream(dev))
        if not ctx.training:
            co.zero_()          # running statistics are constants: dy = scale * dz
        dy = torch.empty_like(yc)
        dres = torch.empty_like(yc) if ctx.has_res else None
        fold = dpass is not None and ctx.has_res and dpass.dtype == yc.dtype
        dadd = dpass.contiguous() if fold else None
        args = (L.ptr(yc), L.ptr(dadd), L.ptr(dout), L.ptr(out), L.ptr(st[2]), None, L.ptr(st[0]), L.ptr(st[1]), L.ptr(co[0]), L.ptr(co[1]), ctx.slope,
                L.ptr(dy), L.ptr(dres), L.dtype_code(yc.dtype), 1, N, C)
        nbytes = yc.numel() * yc.element_size() * (4 + ctx.has_res + fold)
        if mul is not None:
            L.call("p4c_inorm_apply_mul", *args, L.ptr(mul), ctx.mul_rows, ctx.mul_factor, L.stream(dev), alg_bytes=nbytes)
        else:
            L.call("p4c_inorm_apply", *args, L.stream(dev), alg_bytes=nbytes)
        if dpass is not None and not fold:
            dres = dpass if dres is None else dres + dpass
        dg = None if ctx.gdtype is None else dgb[0].to(ctx.gdtype)
        db = None if ctx.gdtype is None else dgb[1].to(ctx.gdtype)
        return (dy, None, dg, db, dres) + (None,) * 10


def batch_norm_act(y, stats, bn: torch.nn.BatchNorm2d, slope: float = 1.0, res=None, res_passthrough=False, mul=None, mul_factor=1.0):
    """``leaky_relu(bn(y) (+ res), slope)`` for a features-last y (B,H,W,C) and a torch.nn.BatchNorm2d module `bn` (its parameters,
    running statistics, momentum, eps and training flag); `stats`: the producer's column sums (conv2d_nhwc(..., want_stats=True)) or
    None.  slope = 1: no activation.  ``res_passthrough``: returns (out, res) -- hand that second tensor to the other consumers of the
    residual operand, and their gradient is added inside this node's backward launch.  ``mul`` (groups, C) fp32 >= 0 with ``mul_factor``:
    the result is multiplied by mul[row group, channel] * mul_factor (row groups = equal consecutive runs of the B*H*W rows: the samples) --
    a channel dropout (Dropout2d) on the block's output, applied by the normalisation passes themselves."""
    L.require_cuda(y)
    if y.dtype not in (torch.bfloat16, torch.float32) or y.shape[-1] % 4 or y.shape[-1] > 1024:
        raise L.P4CError(f"ops_gemm.batch_norm_act: unsupported map {tuple(y.shape)} {y.dtype}")
    training = bn.training or bn.running_mean is None
    if bn.momentum is None and training and bn.track_running_stats:
        # torch: the cumulative moving average, factor 1 / num_batches_tracked -- a per-step value the fused finalize does not take
        raise L.P4CError("ops_gemm.batch_norm_act: BatchNorm2d(momentum=None) (cumulative average) is not served; give a momentum")
    mom = 0.1 if bn.momentum is None else bn.momentum
    nbt = None
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        # incremented by the statistics kernel itself (p4c_bnorm_finalize; round 6 -- a launch of its own before: 42 per model call):
        # on the device, so captured into a HIP graph like the kernels around it (replays advance it)
        nbt = bn.num_batches_tracked
        if nbt.dtype != torch.int64 or nbt.device != y.device:
            raise L.P4CError("ops_gemm.batch_norm_act: num_batches_tracked must be an int64 tensor on the map's device")
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    out, rp = _BatchNormAct.apply(y, stats if training else None, bn.weight, bn.bias, res, rm, rv, training, mom, bn.eps, float(slope), nbt,
                                  bool(res_passthrough), mul, float(mul_factor))
    return (out, rp) if res_passthrough else out


class _UpsampleAdd(torch.autograd.Function):
    """bilinear up-sampling by an integer factor (align_corners = False) of a features-last bf16 map, + skip: csrc/resize.hip"""

    @staticmethod
    def forward(ctx, x, skip, scale):
        xc = x.contiguous()
        B, H, W, C = xc.shape
        sc = None if skip is None else skip.contiguous()
        out = torch.empty(B, H * scale, W * scale, C, dtype=xc.dtype, device=xc.device)
        L.call("p4c_upsample_bilinear_fwd", L.ptr(xc), L.ptr(sc), L.ptr(out), B, H, W, C, scale, L.stream(xc.device),
               alg_bytes=2 * (xc.numel() + out.numel() * (1 + (sc is not None))))
        ctx.geom, ctx.has_skip = (B, H, W, C, scale), skip is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        B, H, W, C, scale = ctx.geom
        dout = dout.contiguous()
        dx = torch.empty(B, H, W, C, dtype=dout.dtype, device=dout.device)
        L.call("p4c_upsample_bilinear_bwd", L.ptr(dout), L.ptr(dx), B, H, W, C, scale, L.stream(dout.device),
               alg_bytes=2 * (dout.numel() + dx.numel()))
        return dx, (dout if ctx.has_skip else None), None


def upsample_add(x: torch.Tensor, skip: Optional[torch.Tensor], scale: int) -> torch.Tensor:
    """``F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=False) (+ skip)`` on features-last (B,H,W,C) bf16 maps"""
    L.require_cuda(x)
    if x.dtype != torch.bfloat16 or x.dim() != 4 or x.shape[-1] % 8 or not (1 <= int(scale) <= 8):
        raise L.P4CError(f"ops_gemm.upsample_add: unsupported map {tuple(x.shape)} {x.dtype} scale {scale}")
    return _UpsampleAdd.apply(x, skip, int(scale))
