"""Autograd wrappers of the row-wise MLP passes (include/py4cast_hip.h: p4c_row_layernorm_fwd/bwd, p4c_row_linear_wgrad).

The GNN models' MLPs (Linear - SiLU - Linear - LayerNorm on 64-feature rows, config/CLI/model/graphlam.yaml:21-22) run over
0.5 M grid nodes and 1-2 M edges per sample.  The forward / input-gradient GEMMs stay with the library (tall-skinny but
efficient); LayerNorm (+ residual) and the weight gradients (64 x K outputs, reduction over millions of rows) are the HIP
kernels of csrc/rows.hip.  No CPU fallback.
"""

import os
from typing import Optional

import torch
import torch.nn.functional as F

from . import _lib as L


class _LayerNormRes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps: float, mask=None):
        L.require_cuda(x)
        x = x.contiguous()
        R, C = x.shape
        mk = (0, 0, 0, 0) if mask is None else tuple(int(v) for v in mask)   # (Hp, Wp, H, W): rows of padding tokens -> zero
        if res is not None:
            res = res.contiguous()
            if res.dtype != x.dtype or res.shape != x.shape:
                raise L.P4CError("row_layer_norm: residual must match x in shape and dtype")
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        out = torch.empty_like(x)
        L.call("p4c_row_layernorm_fwd_masked", L.ptr(x), L.ptr(res), L.ptr(g), L.ptr(b), float(eps), L.ptr(out), R, C,
               L.dtype_code(x.dtype), *mk, L.stream(x.device), alg_bytes=R * C * x.element_size() * (2 + (res is not None)))
        ctx.save_for_backward(x, g)
        ctx.eps, ctx.has_res, ctx.pdtype, ctx.mk = float(eps), res is not None, gamma.dtype, mk
        return out

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        R, C = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgb = torch.empty(2, C, dtype=torch.float32, device=x.device)
        nbytes = L.lib().p4c_row_layernorm_bwd_workspace_bytes(R, C, L.dtype_code(x.dtype))
        ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x.device)
        L.call("p4c_row_layernorm_bwd_masked", L.ptr(dy), L.ptr(x), L.ptr(g), ctx.eps, L.ptr(dx), L.ptr(dgb), L.ptr(dgb[1]), L.ptr(ws), R, C,
               L.dtype_code(x.dtype), *ctx.mk, L.stream(x.device), alg_bytes=3 * R * C * x.element_size())
        return dx, (dy if ctx.has_res else None), dgb[0].to(ctx.pdtype), dgb[1].to(ctx.pdtype), None, None


def row_layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
                   res: Optional[torch.Tensor] = None, mask=None) -> torch.Tensor:
    """``LayerNorm(x) * gamma + beta (+ res)`` on rows: x (R, C).  mask = (Hp, Wp, H, W): the rows are the tokens of (B, Hp, Wp) maps
    whose real extent is (H, W) -- rows of padding tokens come out ZERO and take no gradient (Swin's `F.pad(norm1(x))`)."""
    return _LayerNormRes.apply(x, res, gamma, beta, eps, mask)


def _native_wgrad_ok(x: torch.Tensor, O: int, K: int) -> bool:
    return x.dtype == torch.bfloat16 and O == 64 and K % 16 == 0 and 16 <= K <= 128 and x.shape[0] >= 4096


def _rg_native(x: torch.Tensor, w: torch.Tensor) -> bool:
    """the GNNs' 64-feature projections on the streaming row-GEMM kernel (any row count) instead of a library GEMM: bf16 rows, an fp32
    master weight (a column block of a wider Linear is fine: its row stride is passed on) in the sizes of p4c_row_gemm_supported"""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.float32 and w.dim() == 2 and w.stride(1) == 1 and x.dim() == 2):
        return False
    O, K = w.shape
    if K % 8 or O % 8 or x.shape[0] < 1 or L.diag_switch("P4C_GNN_LIBRARY_GEMM") == "1":
        return False
    lib = L.lib()
    return bool(lib.p4c_row_gemm_supported(K, O) and lib.p4c_row_gemm_supported(O, K))


def _rg_fwd(x, w, b=None):
    return _row_gemm(_rows2d(x, w.shape[1]), w.detach(), False, None if b is None else b.detach().float().contiguous(), w.shape[0])


def _rg_dgrad(dy, w, acc=None):
    """dy W (+ acc, in place when given)"""
    return _row_gemm(_rows2d(dy, w.shape[0]), w.detach(), True, None, w.shape[1], res=acc, out=acc)


class _RowLinear(torch.autograd.Function):
    """y = x W^T + b with W, b fp32 masters and bf16 rows: the row-GEMM kernel (library GEMMs off its sizes) for y and dx,
    p4c_row_linear_wgrad for dW, db."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.native = _rg_native(x, w)
        ctx.has_bias, ctx.pdtype = b is not None, w.dtype
        if ctx.native:
            ctx.save_for_backward(x, w.detach())
            return _rg_fwd(x, w, b)
        wq = w.to(x.dtype)
        ctx.save_for_backward(x, wq)
        return F.linear(x, wq, None if b is None else b.to(x.dtype))

    @staticmethod
    def backward(ctx, dy):
        x, wq = ctx.saved_tensors
        dy = dy.contiguous()
        R, K = x.shape
        O = wq.shape[0]
        dx = (_rg_dgrad(dy, wq) if ctx.native else dy @ wq) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            buf = torch.empty(O * K + O, dtype=torch.float32, device=x.device)
            nbytes = L.lib().p4c_row_linear_wgrad_workspace_bytes(R, K)
            ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x.device)
            L.call("p4c_row_linear_wgrad", L.ptr(dy), L.ptr(x.contiguous()), L.ptr(buf), L.ptr(ws), R, O, K, L.dtype_code(x.dtype),
                   L.stream(x.device), alg_bytes=R * (O + K) * x.element_size())
            dw = buf[: O * K].view(O, K).to(ctx.pdtype)
            db = buf[O * K:].to(ctx.pdtype) if ctx.has_bias else None
        return dx, dw, db


def grad_view(t: Optional[torch.Tensor]):
    """The region of a leaf parameter's .grad that corresponds to t (t = the parameter itself or a basic slice of it): None for
    t = None, False when there is no such buffer (no .grad yet, another dtype or layout)."""
    if t is None:
        return None
    base = L._base(t)      # (the parameter behind a slice, or behind a rollout's per-step stand-in of it)
    if not base.is_leaf:       # (a derived weight: reading .grad of a non-leaf warns)
        return False
    g = base.grad
    if (g is None or g.dtype != torch.float32 or g.shape != base.shape or g.stride() != base.stride()
            or t.dtype != torch.float32):
        return False
    view = g.as_strided(t.shape, t.stride(), t.storage_offset() - base.storage_offset() + g.storage_offset())
    if L.GRAD_SINK_LISTENERS:     # (an exchange that starts inside the backward wants to know which gradients bypass autograd)
        L.grad_sink_taken(view)
    return view


_WCAST = {}   # eager mode: (address, shape, strides, dtype) -> (parameter version, copy in the rows' dtype)


def weight_as(w: torch.Tensor, dtype: torch.dtype, transposed: bool = False) -> torch.Tensor:
    """``w.detach().to(dtype)`` (contiguous; ``transposed``: of ``w.t()``), once per parameter version in eager mode and once per
    HIP-graph capture: the three AR steps of a rollout use the same weight blocks, and a cast is a launch."""
    key = ("wcast_t" if transposed else "wcast", w.data_ptr(), tuple(w.shape), tuple(w.stride()), dtype)
    ver = (L.PARAM_EPOCH[0], w._version)
    capturing = torch.cuda.is_current_stream_capturing()
    cache = _WCAST if not capturing else L.capture_cache()
    if cache is not None:
        hit = cache.get(key)
        if hit is not None and hit[0] == ver and L.owners_alive(hit[2], (w,)):
            return hit[1]
    wq = (w.detach().t() if transposed else w.detach()).to(dtype).contiguous()
    if cache is not None:
        cache[key] = (ver, wq, L.owner_refs((w,)))
    return wq


class _ParamAs(torch.autograd.Function):
    """A parameter in the activation dtype, differentiable: the cast itself is ``weight_as`` -- once per parameter version (and per
    HIP-graph capture), not once per AR step -- and the gradient goes back as one cast.  (What ``torch.autocast``'s weight cache does
    for the reference: the AR steps of a rollout share the casts of the fp32 master weights.)"""

    @staticmethod
    def forward(ctx, w, dtype):
        ctx.dt = w.dtype
        return weight_as(w, dtype).detach()      # a fresh tensor object on the cached storage (the node's output must be its own)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt), None


def param_as(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """``w.to(dtype)`` for a parameter used by several AR steps: cached cast forward, plain cast backward."""
    if w.dtype == dtype:
        return w
    if not w.is_cuda or not w.is_contiguous():
        return w.to(dtype)
    return _ParamAs.apply(w, dtype)


class _RowLinearSink(torch.autograd.Function):
    """``x @ W^T`` on bf16 rows whose weight gradient the backward ADDS into ``gw`` -- the view of the parameter's ``.grad`` that
    corresponds to W -- instead of handing it to autograd.  W is typically a column block of a wider Linear (the sender / receiver
    parts of an edge MLP's first layer): through autograd each backward cost a zero-filled full-size gradient, a copy into its
    slice, a cast and an accumulation (four launches beside the product itself; ~1 700 of them per HiLAM step)."""

    @staticmethod
    def forward(ctx, x, w, gw):
        # w (the live parameter view) is an input so that the node is recorded even when x needs no gradient; its gradient slot
        # returns None: autograd must not also accumulate what the backward adds itself
        ctx.native = _rg_native(x, w)
        ctx.gw = gw
        if ctx.native:
            ctx.save_for_backward(x, w.detach())
            return _rg_fwd(x, w)
        wq = weight_as(w, x.dtype)
        ctx.save_for_backward(x, wq)
        return F.linear(x, wq)

    @staticmethod
    def backward(ctx, dy):
        x, wq = ctx.saved_tensors
        dy = dy.contiguous()
        R, K = x.shape
        O = wq.shape[0]
        dx = (_rg_dgrad(dy, wq) if ctx.native else dy @ wq) if ctx.needs_input_grad[0] else None
        _sink_weight_grad(x, dy, ctx.gw)
        return dx, None, None


def _sink_weight_grad(x, dy, gw):
    """gw += dy^T x: the tall-skinny kernel from 4096 rows, else the library GEMM in the rows' dtype with a promoted add."""
    R, K = x.shape
    O = dy.shape[1]
    if _native_wgrad_ok(x, O, K):
        buf = torch.empty(O * K + O, dtype=torch.float32, device=x.device)
        nbytes = L.lib().p4c_row_linear_wgrad_workspace_bytes(R, K)
        ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x.device)
        L.call("p4c_row_linear_wgrad", L.ptr(dy), L.ptr(x.contiguous()), L.ptr(buf), L.ptr(ws), R, O, K, L.dtype_code(x.dtype),
               L.stream(x.device), alg_bytes=R * (O + K) * x.element_size())
        gw.add_(buf[: O * K].view(O, K))
    else:
        gw.add_(dy.t() @ x)
    L.grad_written(gw)


class _RowLinearMulti(torch.autograd.Function):
    """Several bias-free projections of the SAME rows (x W_1^T, ..., x W_n^T) as one autograd node: the backward accumulates
    dx = sum_i dy_i W_i inside the GEMMs (addmm, beta = 1) instead of leaving n - 1 elementwise additions to autograd, and adds
    every weight gradient into its ``.grad`` view (see _RowLinearSink)."""

    @staticmethod
    def forward(ctx, x, n, *args):
        ws, gws = args[:n], args[n:]
        ctx.native = all(_rg_native(x, w) for w in ws)
        ctx.gws, ctx.n = gws, n
        if ctx.native:
            ctx.save_for_backward(x, *[w.detach() for w in ws])
            return tuple(_rg_fwd(x, w) for w in ws)
        wqs = [weight_as(w, x.dtype) for w in ws]
        ctx.save_for_backward(x, *wqs)
        return tuple(F.linear(x, wq) for wq in wqs)

    @staticmethod
    def backward(ctx, *dys):
        x, *wqs = ctx.saved_tensors
        dx = None
        for dy, wq, gw in zip(dys, wqs, ctx.gws):
            if dy is None:
                continue
            dy = dy.contiguous()
            if ctx.needs_input_grad[0]:
                if ctx.native:
                    dx = _rg_dgrad(dy, wq, acc=dx)        # (the sum over the projections inside the kernel's epilogue)
                else:
                    dx = dy @ wq if dx is None else dx.addmm_(dy, wq)
            _sink_weight_grad(x, dy, gw)
        return (dx, None) + (None,) * (2 * ctx.n)


def row_linear_multi(x: torch.Tensor, weights, grads_in_place: bool = False):
    """``[x @ w.T for w in weights]`` for bf16 rows; with ``grads_in_place`` and gradient buffers for all weights, one autograd
    node (_RowLinearMulti); otherwise one ``row_linear`` each."""
    L.require_cuda(x)
    if grads_in_place and x.dtype == torch.bfloat16 and torch.is_grad_enabled() and all(w.requires_grad for w in weights):
        gws = [grad_view(w) for w in weights]
        if all(g is not None and g is not False for g in gws):
            return _RowLinearMulti.apply(x, len(weights), *weights, *gws)
    return tuple(row_linear(x, w, grads_in_place=grads_in_place) for w in weights)


def row_linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None, grads_in_place: bool = False) -> torch.Tensor:
    """``x @ w.T + b`` for rows x (R, K).  bf16 rows with 64 outputs and K a multiple of 16 (<= 128) take the native weight-gradient
    kernel; everything else (fp32 parity flavour, odd shapes, few rows) is the library's Linear.  ``grads_in_place`` (bias-free
    bf16 case): the weight gradient is accumulated straight into the parameter's ``.grad`` when that buffer exists."""
    L.require_cuda(x)   # no CPU path: the library GEMM below is the GPU library's
    if grads_in_place and b is None and x.dtype == torch.bfloat16 and torch.is_grad_enabled() and w.requires_grad:
        gw = grad_view(w)
        if gw is not None and gw is not False:
            return _RowLinearSink.apply(x, w, gw)
    if _native_wgrad_ok(x, w.shape[0], w.shape[1]):
        return _RowLinear.apply(x, w, b)
    # (few rows -- the upper mesh levels -- or widths off the tall-skinny weight-gradient kernel stay on the library: the row-GEMM
    # weight gradient + its reduction are two launches where the library needs one; measured: HiLAM 44.2 -> 46.8 ms per step with them)
    return F.linear(x, w.to(x.dtype), None if b is None else b.to(x.dtype))


def _rows2d(t: torch.Tensor, C: int) -> torch.Tensor:
    """(..., C) -> (R, C) rows the row-GEMM kernels can address: unit stride inside a row, 16-byte aligned rows (a view when the
    tensor already is that, else one copy)."""
    t2 = t.reshape(-1, C)
    if t2.stride(1) != 1 or t2.stride(0) % 8 or t2.data_ptr() % 16:
        t2 = t2.contiguous()
    return t2


def _row_gemm_mode(x: torch.Tensor, w: torch.Tensor, b) -> str:
    """Which passes of a Linear the native kernels take: "all" (forward, data gradient, weight + bias gradient) or "" (library).  They serve bf16 rows with fp32 master
    parameters, from 1 024 rows (below that the library's ~20 us floor does not matter), in the sizes p4c_row_gemm_supported /
    _wgrad_supported state (include/py4cast_hip.h)."""
    if not (x.is_cuda and x.dtype == torch.bfloat16):
        return ""
    return _row_gemm_mode_rows(x.numel() // max(w.shape[-1], 1), w, b)


def _row_gemm_mode_rows(R: int, w: torch.Tensor, b) -> str:
    """_row_gemm_mode for R bf16 rows on the GPU"""
    if not (w.dtype == torch.float32 and w.dim() == 2 and (b is None or b.dtype == torch.float32)):
        return ""
    O, K = w.shape
    if R < 1024 or K % 8 or O % 8 or _wgrad_chunks(O, K, b is not None) is None:
        return ""
    lib = L.lib()
    if not (lib.p4c_row_gemm_supported(K, O) and lib.p4c_row_gemm_supported(O, K)):
        return ""
    # Measured per shape against the (tuned) library GEMMs, tools/diagnostics/linear_micro.py:
    # * forward / data gradient: every workgroup first lays the weight out as its operand image -- 9 us against the library's 20 us
    #   floor while the image is small against the rows (bytes <= 4 x rows: 32 KiB at 8 192 rows); the 131 KiB images (128 -> 512)
    #   lose outright (34 against 20), and in the SwinUNetR step the 74 KiB ones at 8 192 rows did not pay either (40.6 -> 41.0 ms);
    # * weight + bias gradient: 12-31 us per piece of <= 192 input features against the library's 25-190 us + a 20 us bias GEMM
    #   (64 -> 128 over 32 768 rows: 21 against 96 + 20; 384 -> 96 over 8 192 rows, two pieces: 46 against 26 + 20, the worst case).
    pad = lambda n, m: (n + m - 1) // m * m   # noqa: E731
    steps = lambda k: next(s for s in (16, 32, 48, 64, 96, 128, 192, 256, 384, 512) if s >= k)   # noqa: E731  (the kernel's instantiations)
    image = 2 * max(pad(O, 32) * steps(K), pad(K, 32) * steps(O))
    if image > 80 * 1024 or 4 * R < image:
        return ""
    return "all"


def _row_gemm_ok(x: torch.Tensor, w: torch.Tensor, b) -> bool:
    return _row_gemm_mode(x, w, b) == "all"


_WGRAD_KMAX = 192   # input features per weight-gradient launch (the kernel holds 64 x (K + 1) accumulators per workgroup)


def _wgrad_chunks(O: int, K: int, has_bias: bool):
    """Column ranges of x the weight-gradient kernel takes one launch each (wider inputs -- the 4 x 69-channel pixel blocks of
    SwinUNetR's patch embedding -- go in pieces over strided views of the same rows); None if the layer is out of its range."""
    chunks = [(k0, min(k0 + _WGRAD_KMAX, K)) for k0 in range(0, K, _WGRAD_KMAX)]
    if len(chunks) > 4 or not all(L.lib().p4c_row_gemm_wgrad_supported(O, k1 - k0, int(has_bias and k1 == K)) for k0, k1 in chunks):
        return None
    return chunks


def _row_gemm(x2: torch.Tensor, w: torch.Tensor, transposed: bool, bias, N: int, res=None, act: int = 0, aux=None, out=None) -> torch.Tensor:
    """csrc/rowgemm.hip: y = x2 M^T (+ bias) with the fused epilogue of p4c_row_gemm (act 1: GELU, pre-activation to `aux`; act 2:
    times GELU'(aux); + res).  out: write into this (R, N) tensor (may be `res` itself: accumulate)."""
    R, K = x2.shape
    y = torch.empty(R, N, dtype=x2.dtype, device=x2.device) if out is None else out
    L.call("p4c_row_gemm", L.ptr(x2), x2.stride(0), L.ptr(w), w.stride(0), int(transposed), L.ptr(bias), L.ptr(y), y.stride(0), R, K, N,
           L.ptr(res), 0 if res is None else res.stride(0), int(act), L.ptr(aux), 0 if aux is None else aux.stride(0),
           L.stream(x2.device), alg_bytes=R * (K + N * (1 + (res is not None) + (act != 0))) * 2, alg_flops=2 * R * K * N)
    return y


def _native_wgrad_fn(dy2, x2, O, K, has_bias, wdtype, bdtype):
    """(dW, db) of a Linear from its rows on p4c_row_gemm_wgrad (pieces of <= _WGRAD_KMAX input features)"""
    lib = L.lib()
    R = x2.shape[0]
    parts, db = [], None
    for k0, k1 in _wgrad_chunks(O, K, has_bias):
        kc, ones = k1 - k0, int(has_bias and k1 == K)     # the bias gradient rides with the last piece
        xs = x2[:, k0:k1]
        out = torch.empty(64 * ((O + 63) // 64), 32 * ((kc + ones + 31) // 32), dtype=torch.float32, device=dy2.device)
        ws = torch.empty(max(lib.p4c_row_gemm_wgrad_workspace_bytes(R, O, kc, ones) // 4, 1), dtype=torch.float32, device=dy2.device)
        L.call("p4c_row_gemm_wgrad", L.ptr(dy2), dy2.stride(0), L.ptr(xs), xs.stride(0), L.ptr(out), L.ptr(ws), R, O, kc, ones,
               L.stream(dy2.device), alg_bytes=R * (kc + O) * 2)
        parts.append(out[:O, :kc])
        if ones:
            db = out[:O, kc].to(bdtype)
    dw = (parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)).to(wdtype)
    return dw, db


class _LinearND(torch.autograd.Function):
    """y = x W^T + b on (..., K) tensors of the activation dtype with fp32 master weights.
    bf16 rows in the sizes of SwinUNetR's token layers run on csrc/rowgemm.hip: forward and data gradient as one streaming pass
    each (the weight goes from its fp32 master to the matrix-core operand inside the kernel), weight and bias gradient as one
    reduction over the rows in a fixed order.  Everything else uses library GEMMs for y, dx and dW, with the bias gradient as
    ``ones @ dy`` -- a GEMM too -- instead of a column reduction of dy: under HIP-graph replay at the 512 x 512 sizes the reduction
    route handed back garbage for exactly the Linear biases of SwinUNetR / UNetRPP (tools/diagnostics/nan_probe.py), and a (1 x R)
    GEMM is also the cheaper launch."""

    @staticmethod
    def forward(ctx, x, w, b, res=None, force=False):
        mode = _row_gemm_mode(x, w, b)
        if force and mode != "all" and x.is_cuda and x.dtype == torch.bfloat16 and _wgrad_chunks(w.shape[0], w.shape[1], b is not None) is not None:
            mode = "all"       # the caller has checked the kernels' size limits and wants them whatever the row count (the mesh GNNs)
        ctx.mode = mode
        ctx.has_bias, ctx.wdtype, ctx.bdtype = b is not None, w.dtype, (None if b is None else b.dtype)
        ctx.xshape, ctx.has_res = x.shape, res is not None
        if mode == "all":
            O, K = w.shape
            wc = w.detach() if w.stride(1) == 1 else w.detach().contiguous()
            x2 = _rows2d(x.detach(), K)
            ctx.save_for_backward(x2, wc)
            r2 = None if res is None else _rows2d(res.detach(), O)      # the residual in the product's epilogue
            return _row_gemm(x2, wc, False, None if b is None else b.detach().contiguous(), O, res=r2).view(*x.shape[:-1], O)
        if res is not None:
            raise L.P4CError("_LinearND: a fused residual needs the row-GEMM kernels (callers add it themselves otherwise)")
        wq = w.to(x.dtype)
        ctx.save_for_backward(x, wq)
        return F.linear(x, wq, None if b is None else b.to(x.dtype))

    @staticmethod
    def _native_wgrad(ctx, dy2, x2, O, K):
        return _native_wgrad_fn(dy2, x2, O, K, ctx.has_bias, ctx.wdtype, ctx.bdtype)

    @staticmethod
    def backward(ctx, dy):
        want_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        if ctx.mode == "all":
            x2, wc = ctx.saved_tensors
            O, K = wc.shape
            dy2 = _rows2d(dy, O)
            dx = _row_gemm(dy2, wc, True, None, K).view(ctx.xshape) if ctx.needs_input_grad[0] else None
            dw, db = _LinearND._native_wgrad(ctx, dy2, x2, O, K) if want_w else (None, None)
            return dx, dw, db, (dy if ctx.has_res else None), None
        x, wq = ctx.saved_tensors
        O, K = wq.shape
        dy2 = dy.reshape(-1, O)
        dx = (dy2 @ wq).view(x.shape) if ctx.needs_input_grad[0] else None
        x2 = x.reshape(-1, K)
        dw = (dy2.t() @ x2).to(ctx.wdtype) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            ones = torch.ones(1, dy2.shape[0], dtype=dy2.dtype, device=dy2.device)
            db = (ones @ dy2)[0].to(ctx.bdtype)
        return dx, dw, db, None, None


class _RowMLP(torch.autograd.Function):
    """res + fc2(gelu(fc1(x))) on bf16 token rows as ONE autograd node on the streaming row-GEMM kernels (csrc/rowgemm.hip): GELU in the
    first product's epilogue (the pre-activation h saved), bias + residual in the second's; backward: (dy W2) * GELU'(h) in the
    epilogue of the second layer's data gradient -- no element-wise launch either way (SwinUNETR's two large stages)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, res):
        Hd, K = w1.shape
        O = w2.shape[0]
        x2 = _rows2d(x.detach(), K)
        w1c, w2c = w1.detach().contiguous(), w2.detach().contiguous()
        h = torch.empty(x2.shape[0], Hd, dtype=x2.dtype, device=x2.device)
        g = _row_gemm(x2, w1c, False, b1.detach().contiguous(), Hd, act=1, aux=h)
        r2 = None if res is None else _rows2d(res.detach(), O)
        y = _row_gemm(g, w2c, False, b2.detach().contiguous(), O, res=r2)
        ctx.save_for_backward(x2, g, h, w1c, w2c)
        ctx.xshape, ctx.has_res, ctx.dts = x.shape, res is not None, (w1.dtype, b1.dtype, w2.dtype, b2.dtype)
        ctx.has_bias, ctx.wdtype = True, w1.dtype
        return y.view(*x.shape[:-1], O)

    @staticmethod
    def backward(ctx, dy):
        x2, g, h, w1c, w2c = ctx.saved_tensors
        Hd, K = w1c.shape
        O = w2c.shape[0]
        dy2 = _rows2d(dy, O)
        dh = _row_gemm(dy2, w2c, True, None, Hd, act=2, aux=h)
        ctx.bdtype = ctx.dts[3]
        dw2, db2 = _LinearND._native_wgrad(ctx, dy2, g, O, Hd)
        dx = _row_gemm(dh, w1c, True, None, K).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        ctx.bdtype = ctx.dts[1]
        dw1, db1 = _LinearND._native_wgrad(ctx, dh, x2, Hd, K)
        return dx, dw1.to(ctx.dts[0]), db1, dw2.to(ctx.dts[2]), db2, (dy if ctx.has_res else None)


def row_mlp_gelu_ok(x: torch.Tensor, w1, b1, w2, b2) -> bool:
    return (b1 is not None and b2 is not None and _row_gemm_ok(x, w1, b1)
            and _row_gemm_mode_rows(x.numel() // max(w1.shape[1], 1), w2, b2) == "all")


def row_mlp_gelu(x, w1, b1, w2, b2, res=None) -> torch.Tensor:
    """``F.linear(F.gelu(F.linear(x, w1, b1)), w2, b2) (+ res)`` on the row-GEMM kernels (check row_mlp_gelu_ok first)"""
    return _RowMLP.apply(x, w1, b1, w2, b2, res)


def linear_res(x: torch.Tensor, w: torch.Tensor, b, res: torch.Tensor) -> torch.Tensor:
    """``F.linear(x, w, b) + res`` with the residual in the row-GEMM's epilogue where those kernels serve the layer"""
    if _row_gemm_ok(x, w, b):
        return _LinearND.apply(x, w, b, res)
    return _LinearND.apply(x, w, b) + res


class _AddBias(torch.autograd.Function):
    """x (..., O) + b (O,), with the bias gradient as a ``ones @ dy`` GEMM instead of autograd's sum_to_size column reduction: in HIP-graph
    replays of the 512 x 512 UNetRPP step that reduction returned garbage (1e7 x the eager value) for the token-projection biases once
    the parameters had changed since the capture -- the replay check of trainer.GraphedTrainingStep pins it -- while a GEMM replays
    correctly (same finding as for the Linear biases, _LinearND)."""

    @staticmethod
    def forward(ctx, x, b):
        ctx.bdtype = b.dtype
        return x + b.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        db = None
        if ctx.needs_input_grad[1]:
            dy2 = dy.reshape(-1, dy.shape[-1])
            ones = torch.ones(1, dy2.shape[0], dtype=dy2.dtype, device=dy2.device)
            db = (ones @ dy2)[0].to(ctx.bdtype)
        return dy, db


def add_bias(x: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return _AddBias.apply(x, b)


def linear_nd(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``F.linear`` for activations of any rank with fp32 parameters: the row-GEMM kernels where they apply (csrc/rowgemm.hip),
    else library GEMMs (weight cast once per call, gradients as GEMMs)."""
    O = w.shape[0]
    if O % 8 and w.dim() == 2 and x.is_cuda and x.numel() // max(w.shape[1], 1) >= 65536:
        # an output width off the kernels' 8-feature granularity (SwinUNetR's final 24 -> 60 projection over 524 288 pixels, whose
        # library weight / bias gradients took 290 + 210 us per call): run it with zero rows appended to the weight and slice the
        # result -- two extra copies of the narrow output tensor against three GEMMs at the library's floor
        O8 = (O + 7) // 8 * 8
        wp = F.pad(w, (0, 0, 0, O8 - O))
        bp = None if b is None else F.pad(b, (0, O8 - O))
        if _row_gemm_mode(x, wp, bp) == "all":
            return _LinearND.apply(x, wp, bp)[..., :O]
    return _LinearND.apply(x, w, b)


class _AddLayerNorm(torch.autograd.Function):
    """t = x + add (add: one (N, C) table broadcast over the leading dimension, or None), ln = LayerNorm(t) * gamma + beta -- both
    returned: t is the block's residual, ln its normalised copy (csrc/rows.hip: p4c_row_add_layernorm_fwd / _bwd, rows up to 2 KiB).
    Backward: ONE pass gives dt = LN_backward(d ln) + d t; the table's gradient is its sum over the leading dimension."""

    @staticmethod
    def forward(ctx, x, add, gamma, beta, eps):
        L.require_cuda(x)
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        R = x2.shape[0]
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        a2, arows = None, 0
        if add is not None:
            a2 = weight_as(add, x.dtype).reshape(-1, C)      # cast once per parameter version / capture
            arows = a2.shape[0]
            if R % arows:
                raise L.P4CError("add_layer_norm: the table's rows must divide the rows")
        t = torch.empty_like(x2) if add is not None else x2
        ln = torch.empty_like(x2)
        L.call("p4c_row_add_layernorm_fwd", L.ptr(x2), L.ptr(a2), arows, L.ptr(g), L.ptr(b), float(eps), L.ptr(t if add is not None else None),
               L.ptr(ln), R, C, L.dtype_code(x2.dtype), L.stream(x2.device), alg_bytes=R * C * x2.element_size() * (2 + (add is not None)))
        ctx.save_for_backward(t, g)
        ctx.eps, ctx.shape, ctx.arows = float(eps), x.shape, arows
        ctx.dtypes = (gamma.dtype, None if add is None else add.dtype, None if add is None else add.shape)
        # the table's gradient goes straight into its .grad buffer when that exists (FlatDDP / the trainer allocate them), as ops_gemm's
        # weight gradients do: one pass over dt instead of the tensor library's sum into a fresh tensor + autograd's accumulation
        ctx.asink = None
        if add is not None and add.requires_grad and (R * C) % 4 == 0:
            from . import ops_gemm as G

            if G.GRADS_IN_PLACE:
                gv = grad_view(add)
                if gv is not None and gv is not False and gv.is_contiguous():
                    ctx.asink = gv
        return t.view(x.shape), ln.view(x.shape)

    @staticmethod
    def backward(ctx, dt_in, dln):
        t, g = ctx.saved_tensors
        R, C = t.shape
        dln = dln.reshape(R, C).contiguous()
        extra = None if dt_in is None else dt_in.reshape(R, C).contiguous()
        dt = torch.empty_like(t)
        dgb = torch.empty(2, C, dtype=torch.float32, device=t.device)
        ws = torch.empty(max(L.lib().p4c_row_add_layernorm_bwd_workspace_bytes(R, C) // 4, 1), dtype=torch.float32, device=t.device)
        L.call("p4c_row_add_layernorm_bwd", L.ptr(dln), L.ptr(t), L.ptr(extra), L.ptr(g), ctx.eps, L.ptr(dt), L.ptr(dgb), L.ptr(dgb[1]), L.ptr(ws),
               R, C, L.dtype_code(t.dtype), L.stream(t.device), alg_bytes=R * C * t.element_size() * (3 + (extra is not None)))
        gdt, adt, ashape = ctx.dtypes
        dadd = None
        if ctx.arows and ctx.asink is not None:
            L.call("p4c_sum_leading", L.ptr(dt), L.dtype_code(dt.dtype), R // ctx.arows, ctx.arows * C, L.ptr(ctx.asink), 1, L.stream(t.device),
                   alg_bytes=R * C * dt.element_size() + 8 * ctx.arows * C)
            L.grad_written(ctx.asink)
        elif ctx.arows:
            dadd = dt.view(-1, ctx.arows, C).sum(dim=0, dtype=torch.float32).to(adt).view(ashape)
        return dt.view(ctx.shape), dadd, dgb[0].to(gdt), dgb[1].to(gdt), None


def add_layer_norm_supported(x: torch.Tensor) -> bool:
    nbytes = x.shape[-1] * x.element_size()
    return x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and nbytes % 16 == 0 and nbytes <= 2048


def add_layer_norm(x: torch.Tensor, add: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5):
    """``t = x + add; return t, F.layer_norm(t, (C,), gamma, beta, eps)`` for (..., C) activations, add (N, C) or (1, N, C) or None"""
    return _AddLayerNorm.apply(x, add, gamma, beta, float(eps))
