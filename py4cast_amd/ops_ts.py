"""Autograd wrappers of the tall-skinny products (include/py4cast_hip.h: p4c_ts_gram / p4c_ts_apply) -- the two kernels the efficient
paired attention of UNETR++ is made of, forward and backward (each is the other's adjoint).  Token matrices are (B, heads, N, d)
VIEWS with unit stride in the last dimension and arbitrary strides elsewhere (slices of the qkvv projection, permuted views of
(B, N, C) tensors): they are addressed in place, nothing is made contiguous.  No CPU fallback."""

import ctypes

import torch

from . import _lib as L


def _strides(t: torch.Tensor):
    if t.dim() != 4 or (t.shape[-1] > 1 and t.stride(3) != 1):
        raise L.P4CError("tall-skinny ops take (B, heads, N, d) views with unit stride in the last dimension")
    return t.stride(0), t.stride(1), t.stride(2)


APPLY_MAX_D = 256   # reduction width of one matrix-core apply launch (csrc/tallskinny.hip: p4c_ts_apply_wide_ok)
MAXD = 64   # columns per kernel call (csrc/tallskinny.hip); wider operands (decoder heads of 128 channels) go in column chunks


def _chunks(n):
    return [(i, min(i + MAXD, n)) for i in range(0, n, MAXD)]


def _gram_wide(x, y) -> bool:
    """Served by the matrix-core gram kernel (csrc/tallskinny.hip: gram_mfma_kernel): any width in one launch."""
    if not L.lib().p4c_ts_gram_wide_ok(L.dtype_code(x.dtype), L.dtype_code(y.dtype), x.shape[-1], y.shape[-1]):
        return False
    al = lambda t: all(v % 8 == 0 for v in _strides(t)) and t.data_ptr() % 16 == 0
    return al(x) and al(y)


_I3 = ctypes.c_int * 3
_P3 = ctypes.c_void_p * 3


def reduce_splits(part: torch.Tensor, outs, bias=None, accumulate: bool = False):
    """outs[i] (A, R, len_i) (+)= sum over s of part[a, s, r, segment i] (+ bias, one period of ``bias.numel()`` columns): ``part`` is the
    dense (A, S, R, E) fp32 output of a gram launch, the segments cut its E columns in order (p4c_ts_reduce_splits: one launch, the
    partials added in split order -- no tensor-library reduction, no strided copies of the pieces)."""
    A, S, R = part.shape[:3]
    E = part[0, 0, 0].numel()
    lens = [o.numel() // (A * R) for o in outs]
    for o, n in zip(outs, lens):
        if o.dtype != torch.float32 or not o.is_contiguous() or o.numel() != A * R * n:
            raise L.P4CError("ops_ts.reduce_splits: outputs are dense fp32 (A, R, ...) tensors")
    if not part.is_contiguous() or part.dtype != torch.float32 or sum(lens) != E:
        raise L.P4CError("ops_ts.reduce_splits: partials are a dense fp32 (A, S, R, E) tensor covered by the output segments")
    L.call("p4c_ts_reduce_splits", L.ptr(part), A, S, R, E, len(outs), _I3(*lens, *([0] * (3 - len(lens)))),
           _P3(*[o.data_ptr() for o in outs], *([None] * (3 - len(outs)))), L.ptr(bias), 0 if bias is None else bias.numel(), int(accumulate),
           L.stream(part.device), alg_bytes=4 * A * R * E * (S + 1))
    return outs


_L4 = ctypes.c_int64 * 4
_I4 = ctypes.c_int * 4
_P4 = ctypes.c_void_p * 4


def colsums(jobs):
    """[(x (rows, cols <= 64) fp32 dense, out (cols,) fp32), ...] (<= 4 pairs): out = x.sum(0), all pairs in one launch (p4c_ts_colsums)"""
    n = len(jobs)
    for x, o in jobs:
        if x.dtype != torch.float32 or o.dtype != torch.float32 or x.dim() != 2 or not x.is_contiguous() or not o.is_contiguous() \
                or o.numel() != x.shape[1] or x.shape[1] > 64:
            raise L.P4CError("ops_ts.colsums: dense fp32 (rows, cols <= 64) matrices and (cols,) outputs")
    pad = [None] * (4 - n)
    L.call("p4c_ts_colsums", n, _P4(*[x.data_ptr() for x, _ in jobs], *pad), _L4(*[x.shape[0] for x, _ in jobs], *([0] * (4 - n))),
           _I4(*[x.shape[1] for x, _ in jobs], *([0] * (4 - n))), _P4(*[o.data_ptr() for _, o in jobs], *pad), L.stream(jobs[0][0].device))


def _gram_call(x: torch.Tensor, y: torch.Tensor, out=None) -> torch.Tensor:
    """X^T Y (B, H, d, e) fp32 (into ``out`` when given: a dense fp32 tensor of that many elements)"""
    B, H, N, d = x.shape
    e = y.shape[-1]
    ns = L.lib().p4c_ts_gram_splits(N)
    direct = ns == 1 and out is not None and out.is_contiguous()
    part = out.view(B, 1, H, d, e) if direct else torch.empty(B, ns, H, d, e, dtype=torch.float32, device=x.device)
    xs, ys = _strides(x), _strides(y)
    L.call("p4c_ts_gram", L.ptr(x), L.dtype_code(x.dtype), *xs, L.ptr(y), L.dtype_code(y.dtype), *ys, L.ptr(part), B, H, N, d, e,
           L.stream(x.device), alg_bytes=B * H * N * (d * x.element_size() + e * y.element_size()))
    if direct:
        return out
    if ns == 1 and out is None:
        return part[:, 0]
    if out is None:
        out = torch.empty(B, H, d, e, dtype=torch.float32, device=x.device)
    if (d * e) % 4:
        out.copy_(part.sum(dim=1).view_as(out))
        return out
    reduce_splits(part.view(B, ns, H, d * e), [out.view(B, H, d * e)])
    return out


def _gram_raw(x: torch.Tensor, y: torch.Tensor, out=None) -> torch.Tensor:
    d, e = x.shape[-1], y.shape[-1]
    if (d <= MAXD and e <= MAXD) or _gram_wide(x, y):
        return _gram_call(x, y, out)
    if out is not None:
        raise L.P4CError("ops_ts._gram_raw: an output buffer needs operands one gram launch serves")
    return torch.cat([torch.cat([_gram_call(x[..., i0:i1], y[..., j0:j1]) for j0, j1 in _chunks(e)], dim=-1) for i0, i1 in _chunks(d)], dim=-2)


def _wide(x, out_view, d, e) -> bool:
    """Served by the matrix-core apply kernel (csrc/tallskinny.hip: apply_mfma_kernel), which has no 64-column limit?"""
    if not L.lib().p4c_ts_apply_wide_ok(L.dtype_code(x.dtype), L.dtype_code(out_view.dtype), d, e):
        return False
    al = lambda t: all(v % 8 == 0 for v in _strides(t)) and t.data_ptr() % 16 == 0
    return al(x) and al(out_view)


def _stored_transposed(m: torch.Tensor):
    """m (B,H,d,e) fp32 that is the transposed VIEW of a contiguous (B,H,e,d) tensor -> that tensor, else None"""
    if m.dtype == torch.float32 and m.dim() == 4 and not m.is_contiguous() and m.shape[-1] > 1 and m.shape[-2] > 1:
        t = m.transpose(-1, -2)
        if t.is_contiguous():
            return t
    return None


def _apply_call(x, m, ov, B, H, N, d, e, accumulate, m_gs=None):
    """out (+)= x @ m on the matrix-core kernel; a transposed view of a stored matrix is read as stored (p4c_ts_apply_mt), anything
    else is made contiguous first.  m_gs: m is a block of d consecutive ROWS of each group's dense fp32 matrix, the groups m_gs
    elements apart (a chunk of a wider matrix: read in place)."""
    args_x = (L.ptr(x), L.dtype_code(x.dtype), *_strides(x))
    args_o = (L.ptr(ov), L.dtype_code(ov.dtype), *_strides(ov))
    nbytes = B * H * N * (d * x.element_size() + e * ov.element_size() * (1 + int(accumulate)))
    if m_gs is not None:
        if m.dtype != torch.float32 or m.stride(-1) != 1 or m.stride(-2) != e:
            raise L.P4CError("ops_ts._apply_call: a row block of dense fp32 matrices expected")
        L.call("p4c_ts_apply", *args_x, L.ptr(m), m_gs, *args_o, B, H, N, d, e, int(accumulate), L.stream(x.device), alg_bytes=nbytes)
        return
    mt = _stored_transposed(m)
    if mt is not None and L.lib().p4c_ts_apply_mt_ok(*args_x, L.ptr(mt), d * e, *args_o, d, e):
        L.call("p4c_ts_apply_mt", *args_x, L.ptr(mt), d * e, *args_o, B, H, N, d, e, int(accumulate), L.stream(x.device), alg_bytes=nbytes)
        return
    m = m.float().contiguous()
    L.call("p4c_ts_apply", *args_x, L.ptr(m), d * e, *args_o, B, H, N, d, e, int(accumulate), L.stream(x.device), alg_bytes=nbytes)


def _apply_raw(x: torch.Tensor, m: torch.Tensor, dtype) -> torch.Tensor:
    """x (B,H,N,d) @ m (B,H,d,e) fp32 -> (B,H,N,e) view of a fresh (B,N,H,e) tensor (i.e. laid out as (B, N, H*e) tokens)."""
    B, H, N, d = x.shape
    e = m.shape[-1]
    out = torch.empty(B, N, H, e, dtype=dtype, device=x.device)
    ov = out.permute(0, 2, 1, 3)
    m = m.float()
    if _wide(x, ov, d, e):      # matrix-core kernel: any width in one launch
        _apply_call(x, m, ov, B, H, N, d, e, False)
        return ov
    for j0, j1 in _chunks(e):
        oj = ov[..., j0:j1]
        for k, (i0, i1) in enumerate(_chunks(d)):
            xi, mi = x[..., i0:i1], m[:, :, i0:i1, j0:j1].contiguous()
            L.call("p4c_ts_apply", L.ptr(xi), L.dtype_code(x.dtype), *_strides(xi), L.ptr(mi), (i1 - i0) * (j1 - j0), L.ptr(oj),
                   L.dtype_code(dtype), *_strides(oj), B, H, N, i1 - i0, j1 - j0, int(k > 0), L.stream(x.device),
                   alg_bytes=B * H * N * ((i1 - i0) * x.element_size() + (j1 - j0) * out.element_size() * (1 + (k > 0))))
    return ov


class _Gram(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        L.require_cuda(x, y)
        ctx.save_for_backward(x, y)
        return _gram_raw(x, y)

    @staticmethod
    def backward(ctx, dc):
        x, y = ctx.saved_tensors
        dx = _apply_raw(y, dc.transpose(-1, -2), x.dtype) if ctx.needs_input_grad[0] else None    # dX = Y dC^T
        dy = _apply_raw(x, dc, y.dtype) if ctx.needs_input_grad[1] else None                      # dY = X dC
        return dx, dy


class _Apply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, m, out_dtype):
        L.require_cuda(x, m)
        ctx.save_for_backward(x, m)
        return _apply_raw(x, m, out_dtype)

    @staticmethod
    def backward(ctx, dout):
        x, m = ctx.saved_tensors
        if dout.stride(3) != 1:
            dout = dout.contiguous()
        dx = _apply_raw(dout, m.transpose(-1, -2), x.dtype) if ctx.needs_input_grad[0] else None  # dX = dO M^T
        dm = _gram_raw(x, dout).to(m.dtype) if ctx.needs_input_grad[1] else None                  # dM = X^T dO
        return dx, dm, None


def gram(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """(B,H,N,d), (B,H,N,e) -> X^T Y (B,H,d,e) fp32: the reduction over the tokens (d, e multiples of 4, <= 64)."""
    return _Gram.apply(x, y)


def apply(x: torch.Tensor, m: torch.Tensor, out_dtype=None) -> torch.Tensor:
    """(B,H,N,d) @ (B,H,d,e) -> (B,H,N,e), stored token-major ((B,N,H,e) memory): ``.permute(0,2,1,3).reshape(B,N,H*e)`` is a view."""
    return _Apply.apply(x, m, out_dtype or x.dtype)


def _apply_into(out_view, x, m, accumulate):
    """out_view (B,H,N,e) (+)= x (B,H,N,d) @ m (B,H,d,e): p4c_ts_apply into an existing token-major buffer (d, e <= 64)."""
    B, H, N, d = x.shape
    e = m.shape[-1]
    _apply_call(x, m.float(), out_view, B, H, N, d, e, accumulate)


class _GramNorms(torch.autograd.Function):
    """G = X^T Y together with the column sums of squares nx2 = sum_n X_ni^2, ny2 = sum_n Y_nj^2 -- what EPA needs of q and k -- from
    ONE pass over the two token matrices (p4c_ts_gram_norms) instead of three gram launches; backward dX = Y dG^T + 2 X diag(dnx2),
    dY = X dG + 2 Y diag(dny2): two apply launches per operand, the second accumulating into the first's output."""

    @staticmethod
    def forward(ctx, x, y):
        L.require_cuda(x, y)
        B, H, N, d = x.shape
        e = y.shape[-1]
        ns = L.lib().p4c_ts_gram_splits(N)
        part = torch.empty(B, ns, H, d * e + d + e, dtype=torch.float32, device=x.device)
        L.call("p4c_ts_gram_norms", L.ptr(x), L.dtype_code(x.dtype), *_strides(x), L.ptr(y), L.dtype_code(y.dtype), *_strides(y), L.ptr(part), B, H,
               N, d, e, L.stream(x.device), alg_bytes=B * H * N * (d * x.element_size() + e * y.element_size()))
        ctx.save_for_backward(x, y)
        if ns == 1 or d % 4 or e % 4:
            tot = part.sum(dim=1) if ns > 1 else part[:, 0]
            return tot[..., : d * e].reshape(B, H, d, e), tot[..., d * e : d * e + d], tot[..., d * e + d :]
        G, nx2, ny2 = (torch.empty(B, H, n, dtype=torch.float32, device=x.device) for n in (d * e, d, e))
        reduce_splits(part, [G, nx2, ny2])
        return G.view(B, H, d, e), nx2, ny2

    @staticmethod
    def backward(ctx, dG, dnx2, dny2):
        x, y = ctx.saved_tensors
        B, H, N, d = x.shape
        e = y.shape[-1]
        dx = dy = None
        if ctx.needs_input_grad[0]:
            out = torch.empty(B, N, H, d, dtype=x.dtype, device=x.device)
            dx = out.permute(0, 2, 1, 3)
            have = False
            if dG is not None:
                _apply_into(dx, y, dG.transpose(-1, -2), False)
                have = True
            if dnx2 is not None:
                _apply_into(dx, x, torch.diag_embed(2.0 * dnx2.float()), have)
                have = True
            if not have:
                dx = None
        if ctx.needs_input_grad[1]:
            out = torch.empty(B, N, H, e, dtype=y.dtype, device=y.device)
            dy = out.permute(0, 2, 1, 3)
            have = False
            if dG is not None:
                _apply_into(dy, x, dG, False)
                have = True
            if dny2 is not None:
                _apply_into(dy, y, torch.diag_embed(2.0 * dny2.float()), have)
                have = True
            if not have:
                dy = None
        return dx, dy


def gram_norms(x: torch.Tensor, y: torch.Tensor):
    """(X^T Y (B,H,d,e), column sums of squares of X (B,H,d) and of Y (B,H,e)), fp32, from one pass (bf16 token matrices, d, e <= 64)."""
    return _GramNorms.apply(x, y)


def spatial_fused_ok(q: torch.Tensor, p: int) -> bool:
    """Can ``epa_spatial`` serve these token matrices (bf16, matrix-core apply, p <= 64, 16-byte aligned rows)?"""
    d = q.shape[-1]
    if q.dtype != torch.bfloat16 or p > 64 or not L.lib().p4c_ts_apply_wide_ok(L.dtype_code(q.dtype), L.dtype_code(q.dtype), d, p):
        return False
    return bool(L.lib().p4c_ts_apply_wide_ok(L.dtype_code(q.dtype), L.dtype_code(q.dtype), p, d)) and \
        all(v % 8 == 0 for v in _strides(q)) and q.data_ptr() % 16 == 0


def _apply_softmax(x, m, epi, s=None):
    B, H, N, d = x.shape
    e = m.shape[-1]
    out = torch.empty(B, N, H, e, dtype=x.dtype, device=x.device)
    ov = out.permute(0, 2, 1, 3)
    m = m.float().contiguous()
    ss = _strides(s) if s is not None else (0, 0, 0)
    L.call("p4c_ts_apply_softmax", L.ptr(x), *_strides(x), L.ptr(m), d * e, L.ptr(ov), *_strides(ov), B, H, N, d, e, epi,
           L.ptr(s) if s is not None else None, *ss, L.stream(x.device),
           alg_bytes=B * H * N * (d + e * (2 if s is not None else 1)) * x.element_size())
    return ov


class _EpaSpatial(torch.autograd.Function):
    """The spatial branch of EPA as one node: x_sa = softmax(q Mq) VP^T with the row softmax -- and, backward, the softmax adjoint -- in
    the epilogue of the apply that produces its argument (p4c_ts_apply_softmax): S is written once and read once forward, the N x p
    logits and their gradient never exist in memory (as separate ops: two more passes over (B, heads, N, p) each way)."""

    @staticmethod
    def forward(ctx, q, Mq, VPt):
        L.require_cuda(q, Mq, VPt)
        S = _apply_softmax(q, Mq, 1)                       # (B,h,N,p), token-major
        x = _apply_raw(S, VPt, q.dtype)                    # (B,h,N,d)
        ctx.save_for_backward(q, Mq, VPt, S)
        return x

    @staticmethod
    def backward(ctx, dx):
        q, Mq, VPt, S = ctx.saved_tensors
        if dx.stride(3) != 1 or any(v % 8 for v in _strides(dx)) or dx.data_ptr() % 16:
            dx = dx.contiguous()
        dL = _apply_softmax(dx, VPt.transpose(-1, -2), 2, S)        # dS = dx VP, then the softmax adjoint in the epilogue
        dVPt = _gram_raw(S, dx).to(VPt.dtype) if ctx.needs_input_grad[2] else None
        dq = _apply_raw(dL, Mq.transpose(-1, -2), q.dtype) if ctx.needs_input_grad[0] else None
        dMq = _gram_raw(q, dL).to(Mq.dtype) if ctx.needs_input_grad[1] else None
        return dq, dMq, dVPt


def epa_spatial(q: torch.Tensor, Mq: torch.Tensor, VPt: torch.Tensor) -> torch.Tensor:
    """softmax(q (B,h,N,d) @ Mq (B,h,d,p), dim=-1) @ VPt (B,h,p,d) -> (B,h,N,d) token-major; see ``spatial_fused_ok``."""
    return _EpaSpatial.apply(q, Mq, VPt)


class _EpaSmall(torch.autograd.Function):
    """The small matrices of an EPA block as one native launch each way (p4c_epa_small_fwd / _bwd; see csrc/tallskinny.hip):
    (G, Gq, Gk (B,h,d,d) fp32, KP (B,h,d,p) fp32, t1, t2 (h,1,1)) -> At = softmax(t1 G / (nq nk^T))^T (B,h,d,d), Mq = t2 KP / nq (B,h,d,p)."""

    @staticmethod
    def forward(ctx, G, Gq, Gk, KP, t1, t2):
        L.require_cuda(G, KP)
        B, H, d, _ = G.shape
        p = KP.shape[-1]
        ctx.diag = int(Gq.dim() == 3)     # Gq / Gk given as their diagonals (B,H,d): the squared column norms of q and k
        G, Gq, Gk, KP = (t.detach().float().contiguous() for t in (G, Gq, Gk, KP))
        t1f, t2f = t1.detach().float().reshape(-1).contiguous(), t2.detach().float().reshape(-1).contiguous()
        At = torch.empty(B, H, d, d, dtype=torch.float32, device=G.device)
        Mq = torch.empty(B, H, d, p, dtype=torch.float32, device=G.device)
        nq = torch.empty(2, B, H, d, dtype=torch.float32, device=G.device)
        L.call("p4c_epa_small_fwd", L.ptr(G), L.ptr(Gq), L.ptr(Gk), L.ptr(KP), L.ptr(t1f), L.ptr(t2f), L.ptr(At), L.ptr(Mq), L.ptr(nq[0]),
               L.ptr(nq[1]), B, H, d, p, ctx.diag, L.stream(G.device))
        ctx.save_for_backward(G, Gq, Gk, KP, t1f, t2f, At, nq)
        ctx.tshape, ctx.tdtype = t1.shape, t1.dtype
        return At, Mq

    @staticmethod
    def backward(ctx, dAt, dMq):
        G, Gq, Gk, KP, t1f, t2f, At, nq = ctx.saved_tensors
        B, H, d, _ = G.shape
        p = KP.shape[-1]
        dAt = torch.zeros_like(At) if dAt is None else dAt.float().contiguous()
        dMq = torch.zeros_like(KP) if dMq is None else dMq.float().contiguous()
        dG = torch.empty(B, H, d, d, dtype=torch.float32, device=G.device)
        dd = torch.empty((2, B, H, d) if ctx.diag else (2, B, H, d, d), dtype=torch.float32, device=G.device)
        dKP = torch.empty_like(KP)
        dt = torch.empty(2, B, H, dtype=torch.float32, device=G.device)
        L.call("p4c_epa_small_bwd", L.ptr(G), L.ptr(Gq), L.ptr(Gk), L.ptr(KP), L.ptr(t1f), L.ptr(t2f), L.ptr(At), L.ptr(nq[0]), L.ptr(nq[1]),
               L.ptr(dAt), L.ptr(dMq), L.ptr(dG), L.ptr(dd[0]), L.ptr(dd[1]), L.ptr(dKP), L.ptr(dt[0]), L.ptr(dt[1]), B, H, d, p, ctx.diag,
               L.stream(G.device))
        dts = dt.sum(dim=1)                                             # (2, H): over the samples, fixed order
        return dG, dd[0], dd[1], dKP, dts[0].view(ctx.tshape).to(ctx.tdtype), dts[1].view(ctx.tshape).to(ctx.tdtype)


def epa_small(G, Gq, Gk, KP, t1, t2):
    """(At, Mq) of an EPA block from its gram matrices, the token projection KP and the two temperatures (see _EpaSmall)."""
    return _EpaSmall.apply(G, Gq, Gk, KP, t1, t2)


_TWO_EYE = {}


def _two_eye(d: int, device) -> torch.Tensor:
    """2 I (d x d), made once per device -- and once per HIP-graph capture (a tensor made while capturing lives in the graph's pool and
    holds nothing until a replay has run: it must not leak into the process-wide cache)."""
    key = ("two_eye", d, str(device))
    capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
    cache = L.capture_cache() if capturing else _TWO_EYE
    if cache is None:
        return 2.0 * torch.eye(d, dtype=torch.float32, device=device)
    if key not in cache:
        cache[key] = 2.0 * torch.eye(d, dtype=torch.float32, device=device)
    return cache[key]


class _EpaCore(torch.autograd.Function):
    """The whole efficient paired attention between the qkvv projection and the two output projections as ONE autograd node
    (bf16 flavour, d <= 128, p <= 64): qkvv (B, N, 4, heads, d) -> (x_sa, x_ca), both (B, heads, N, d) token-major.  The pieces are the
    kernels of this module (gram with norms, the small-matrix kernel, apply, apply with the softmax epilogues) and the library GEMM of
    the token-axis projection E; what the node adds is the BACKWARD's bookkeeping: dq, dk, dv_ca, dv_sa are written straight into one
    (B, N, 4, heads, d) gradient by accumulating applies -- through separate nodes autograd added q's and k's two contributions each
    with its own kernel and a fifth node copied the four gradients into that buffer (four copies + two additions per block)."""

    @staticmethod
    def forward(ctx, qkvv, W, bias, t1, t2):
        L.require_cuda(qkvv, W)
        B, N, _, H, d = qkvv.shape
        C, p, dt = H * d, W.shape[0], qkvv.dtype
        q, k, vca, vsa = (qkvv[:, :, i].permute(0, 2, 1, 3) for i in range(4))
        # q^T k with the squared column norms of q and k: one gram pass, one launch that sums the splits into the three pieces
        ns = L.lib().p4c_ts_gram_splits(N)
        part = torch.empty(B, ns, H, d * d + 2 * d, dtype=torch.float32, device=qkvv.device)
        L.call("p4c_ts_gram_norms", L.ptr(q), L.dtype_code(dt), *_strides(q), L.ptr(k), L.dtype_code(dt), *_strides(k), L.ptr(part), B, H, N, d, d,
               L.stream(qkvv.device), alg_bytes=B * H * N * 2 * d * q.element_size())
        G = torch.empty(B, H, d, d, dtype=torch.float32, device=qkvv.device)
        nq2, nk2 = (torch.empty(B, H, d, dtype=torch.float32, device=qkvv.device) for _ in range(2))   # |q columns|^2, |k columns|^2
        reduce_splits(part, [G, nq2, nk2])
        # token-axis projection of k and v_sa (shared weights E = F): KP[b,h] = k[b,h]^T W^T + bias, (d x p) per head
        from .ops_rows import weight_as

        proj = torch.empty(2, B, C, p, dtype=torch.float32, device=qkvv.device)      # KP and VP are its two halves
        native = _token_proj_native(qkvv, p)
        if native:
            # round 6: the projection is a gram product of the token matrices with W^T (N x p), over ALL channels of a sample at once --
            # group (k | v_sa, sample), k and v_sa read in place inside qkvv as (N x C) matrices with row stride 4 C, the shared
            # operand with group strides 0; the splits' partials, the cast and the bias are one more launch.  (Rounds 3-5: a strided
            # gather of k / v_sa into (2, B, N, C), its transposed copy for the library's batched GEMM, the GEMM, a cast and an addition:
            # five launches and five passes over 16 MB at the first stage.)
            Wm = weight_as(W, dt, transposed=True)                           # W^T (N, p), once per parameter version
            kv = None
            part = _gram_partial(_kv_view(qkvv), Wm.view(1, 1, N, p).expand(2, B, N, p))       # (2, splits, B, C, p)
            reduce_splits(part.view(2, part.shape[1], B, C * p), [proj], bias=bias.detach().float().contiguous())
        else:
            Wm = weight_as(W, dt)                                                # (p, N)
            kv = qkvv[:, :, 1::2].permute(2, 0, 1, 3, 4).reshape(2, B, N, C)      # k and v_sa token-major, ONE strided copy: (2,B,N,C)
            proj = (kv.transpose(-1, -2) @ Wm.t()).float() + bias.float()        # (2,B,C,p)
        KP, VP = proj[0].view(B, H, d, p), proj[1].view(B, H, d, p)
        t1f, t2f = t1.detach().float().reshape(-1).contiguous(), t2.detach().float().reshape(-1).contiguous()
        At = torch.empty(B, H, d, d, dtype=torch.float32, device=qkvv.device)
        Mq = torch.empty(B, H, d, p, dtype=torch.float32, device=qkvv.device)
        nrm = torch.empty(2, B, H, d, dtype=torch.float32, device=qkvv.device)
        L.call("p4c_epa_small_fwd", L.ptr(G), L.ptr(nq2), L.ptr(nk2), L.ptr(KP), L.ptr(t1f), L.ptr(t2f), L.ptr(At), L.ptr(Mq), L.ptr(nrm[0]),
               L.ptr(nrm[1]), B, H, d, p, 1, L.stream(qkvv.device))
        x_ca = _apply_raw(vca, At, dt)
        S = _apply_softmax(q, Mq, 1)
        x_sa = _apply_raw(S, VP.transpose(-1, -2), dt)
        ctx.native_proj = native
        from .ops_gemm import _sink

        ctx.wsink = _sink(W, None) if native else None     # (the weight gradient goes straight into E.weight.grad, as ops_gemm's do)
        ctx.save_for_backward(qkvv, Wm, kv, G, nq2, nk2, KP, VP, t1f, t2f, At, Mq, nrm, S)
        ctx.meta = (W.dtype, bias.dtype, t1.shape, t1.dtype)
        return x_sa, x_ca

    @staticmethod
    def backward(ctx, dx_sa, dx_ca):
        qkvv, Wm, kv, G, nq2, nk2, KP, VP, t1f, t2f, At, Mq, nrm, S = ctx.saved_tensors
        wdt, bdt, tshape, tdt = ctx.meta
        B, N, _, H, d = qkvv.shape
        C, p, dt = H * d, KP.shape[-1], qkvv.dtype
        dev = qkvv.device
        q, k, vca, vsa = (qkvv[:, :, i].permute(0, 2, 1, 3) for i in range(4))
        ok = lambda t: t.stride(3) == 1 and all(v % 8 == 0 for v in _strides(t)) and t.data_ptr() % 16 == 0   # noqa: E731
        dx_sa = dx_sa if ok(dx_sa) else dx_sa.contiguous()
        dx_ca = dx_ca if ok(dx_ca) else dx_ca.contiguous()
        dqkvv = torch.empty_like(qkvv)
        dq, dk, dvca, dvsa = (dqkvv[:, :, i].permute(0, 2, 1, 3) for i in range(4))
        g = torch.empty(2, B, C, p, dtype=torch.float32, device=dev)         # (dKP, dVP): the gradient of the token-axis projection
        dKP, dVP = g[0].view(B, H, d, p), g[1].view(B, H, d, p)
        # channel branch x_ca = v_ca At
        _apply_into(dvca, dx_ca, At.transpose(-1, -2), False)
        dAt = _gram_raw(vca, dx_ca)
        # spatial branch x_sa = softmax(q Mq) VP^T
        dL = _apply_softmax(dx_sa, VP, 2, S)                                 # dS = dx VP, softmax adjoint in the epilogue
        _gram_raw(dx_sa, S, out=dVP)                                         # d(VP^T) = S^T dx, i.e. dVP = dx^T S (B,H,d,p)
        _apply_into(dq, dL, Mq.transpose(-1, -2), False)                     # dq, first contribution
        dMq = _gram_raw(q, dL)
        # small matrices
        dG = torch.empty(B, H, d, d, dtype=torch.float32, device=dev)
        dn = torch.empty(2, B, H, d, dtype=torch.float32, device=dev)
        dtp = torch.empty(2, B, H, dtype=torch.float32, device=dev)
        L.call("p4c_epa_small_bwd", L.ptr(G), L.ptr(nq2), L.ptr(nk2), L.ptr(KP), L.ptr(t1f), L.ptr(t2f), L.ptr(At), L.ptr(nrm[0]), L.ptr(nrm[1]),
               L.ptr(dAt.float().contiguous()), L.ptr(dMq.float().contiguous()), L.ptr(dG), L.ptr(dn[0]), L.ptr(dn[1]), L.ptr(dKP), L.ptr(dtp[0]),
               L.ptr(dtp[1]), B, H, d, p, 1, L.stream(dev))
        # the temperatures' gradients (sums over the samples) and the bias gradient of the token-axis projection proj = kv^T W^T + bias
        # (sum over k | v_sa, samples and channels): one launch for the three column sums
        dts = torch.empty(2, H, dtype=torch.float32, device=dev)
        dbias = torch.empty(p, dtype=torch.float32, device=dev)
        colsums([(dtp[0], dts[0]), (dtp[1], dts[1]), (g.view(2 * B * C, p), dbias)])
        dbias = dbias.to(bdt)
        dW = None
        if ctx.native_proj:
            # the adjoints as tall-skinny products on the operands in place, groups = (k | v_sa, sample), all heads' channels at once:
            # (dk | dv_sa)[g] (N x C) = W^T (N x p) g[g]^T (p x C) written straight into dqkvv (first contributions), and
            # dW^T (N x p) = sum over the groups of X[g] (N x C) g[g] (C x p): one apply into per-group fp32 products, one launch that adds
            # the four and transposes them into (p x N) -- into E.weight.grad itself when that buffer exists.  No bf16 copy of g, no
            # (2, B, N, C) gradient tensor, no scatter copy, no library GEMM.
            Wv = Wm.view(1, 1, N, p).expand(2, B, N, p)
            _apply_call(Wv, g.transpose(-1, -2), _kv_view(dqkvv), 2, B, N, p, C, False)
            prod = torch.empty(2, B, N, p, dtype=torch.float32, device=dev)
            Xkv = _kv_view(qkvv)
            for c0 in range(0, C, APPLY_MAX_D):       # (the apply kernel reduces over <= 256 columns per launch: wider rows in chunks)
                c1 = min(c0 + APPLY_MAX_D, C)
                _apply_call(Xkv[..., c0:c1], g[:, :, c0:c1], prod, 2, B, N, c1 - c0, p, c0 > 0, m_gs=C * p)
            if ctx.wsink is not None:
                out, acc = ctx.wsink[0], 1
            else:
                out, acc = torch.empty(p, N, dtype=torch.float32, device=dev), 0
            L.call("p4c_ts_reduce_transpose", L.ptr(prod), 2 * B, N, p, L.ptr(out), acc, L.stream(dev), alg_bytes=4 * N * p * (2 * B + 1 + acc))
            if acc:
                L.grad_written(out)
            else:
                dW = out.to(wdt)
        else:
            g16 = g.to(dt)
            dkv = (g16 @ Wm).transpose(-1, -2)                                           # (2,B,N,C) view of (2,B,C,N)
            dW = torch.bmm(g16.reshape(2 * B, C, p).transpose(1, 2), kv.reshape(2 * B, N, C).transpose(1, 2)).sum(dim=0).to(wdt)   # (p,N)
            dqkvv[:, :, 1::2].copy_(dkv.reshape(2, B, N, H, d).permute(1, 2, 0, 3, 4))    # dk (first contribution) and dv_sa: one copy
        # q^T k and the norms: dq += k dG^T + 2 q diag(dnq2),  dk += q dG + 2 k diag(dnk2)
        D = dn.unsqueeze(-1) * _two_eye(d, dn.device)                                # (2,B,H,d,d) = 2 diag(dn), one launch
        _apply_into(dq, k, dG.transpose(-1, -2), True)
        _apply_into(dq, q, D[0], True)
        _apply_into(dk, q, dG, True)
        _apply_into(dk, k, D[1], True)
        return dqkvv, dW, dbias, dts[0].view(tshape).to(tdt), dts[1].view(tshape).to(tdt)


def _kv_view(qkvv: torch.Tensor) -> torch.Tensor:
    """k and v_sa inside a (B, N, 4, heads, d) tensor as (2, B, N, C) token matrices IN PLACE: group strides (2 C, 4 N C), row stride 4 C"""
    B, N, _, H, d = qkvv.shape
    C = H * d
    if not qkvv.is_contiguous():
        raise L.P4CError("ops_ts._kv_view: the qkvv tensor must be dense")
    return qkvv.as_strided((2, B, N, C), (2 * C, 4 * N * C, 4 * C, 1), qkvv.storage_offset() + C)


def _gram_partial(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """the per-split partials (B, splits, H, d, e) of X^T Y (p4c_ts_gram): the caller sums them where it wants the result"""
    B, H, N, d = x.shape
    e = y.shape[-1]
    ns = L.lib().p4c_ts_gram_splits(N)
    part = torch.empty(B, ns, H, d, e, dtype=torch.float32, device=x.device)
    L.call("p4c_ts_gram", L.ptr(x), L.dtype_code(x.dtype), *_strides(x), L.ptr(y), L.dtype_code(y.dtype), *_strides(y), L.ptr(part), B, H, N, d, e,
           L.stream(x.device), alg_bytes=B * H * N * d * x.element_size() + N * e * y.element_size())
    return part


def _token_proj_native(qkvv: torch.Tensor, p: int) -> bool:
    """EPA's token-axis projection and its adjoints on the tall-skinny kernels, k / v_sa read in place (see _EpaCore): bf16, all heads'
    channels of a token as one row (the weight gradient's apply in chunks of 256 channels).  ``P4C_EPA_LIB_PROJ=1`` (diagnostic
    library) keeps the gather + library GEMM route of rounds 3-5 for A/B runs."""
    B, N, _, H, d = qkvv.shape
    C = H * d
    if L.diag_switch("P4C_EPA_LIB_PROJ") == "1" or C % 8 or p % 8 or p > 64 or (C > APPLY_MAX_D and C % APPLY_MAX_D):
        return False
    lib, bf, f32 = L.lib(), L.dtype_code(torch.bfloat16), L.dtype_code(torch.float32)
    return bool(lib.p4c_ts_gram_wide_ok(bf, bf, C, p) and lib.p4c_ts_apply_wide_ok(bf, bf, p, C)
                and lib.p4c_ts_apply_wide_ok(bf, f32, min(C, APPLY_MAX_D), p))


def epa_core_ok(qkvv: torch.Tensor, p: int) -> bool:
    """bf16 qkvv (B, N, 4, heads, d) with d a multiple of 8 up to 128 (8, 16, 32, 64, 128), p a multiple of 8 up to 64 and 16-byte aligned
    head rows."""
    if qkvv.dim() != 5 or qkvv.dtype != torch.bfloat16 or not qkvv.is_contiguous():
        return False
    d = qkvv.shape[-1]
    return d % 8 == 0 and p % 8 == 0 and d <= 128 and p <= 64 and 64 % (d // 8) == 0 and spatial_fused_ok(qkvv[:, :, 0].permute(0, 2, 1, 3), p)


def epa_core(qkvv, W, bias, t1, t2):
    """(x_sa, x_ca) of an EPA block from its qkvv projection, the token-axis Linear E (weight (p, N), bias) and the temperatures."""
    return _EpaCore.apply(qkvv, W, bias, t1, t2)


class _MergePublished(torch.autograd.Function):
    """``x_sa.permute(0, 3, 1, 2).reshape(B, N, C)`` of the published EPA code on the token-major (B, heads, N, d) view the apply kernels
    produce, and its adjoint, as tiled transposes (p4c_ts_merge_published) instead of the tensor library's strided gathers; the gradient
    comes back token-major, i.e. as the (B, heads, N, d) view with unit stride the backward kernels of the block read in place."""

    @staticmethod
    def forward(ctx, x_sa):
        B, H, N, d = x_sa.shape
        ctx.geom = (B, H, N, d)
        out = torch.empty(B, N, H * d, dtype=x_sa.dtype, device=x_sa.device)
        L.call("p4c_ts_merge_published", L.ptr(x_sa), L.ptr(out), B, N, H, d, 0, L.stream(x_sa.device), alg_bytes=2 * out.numel() * out.element_size())
        return out

    @staticmethod
    def backward(ctx, dout):
        B, H, N, d = ctx.geom
        dout = dout.contiguous()
        dx = torch.empty(B, N, H, d, dtype=dout.dtype, device=dout.device)
        L.call("p4c_ts_merge_published", L.ptr(dout), L.ptr(dx), B, N, H, d, 1, L.stream(dout.device), alg_bytes=2 * dx.numel() * dx.element_size())
        return dx.permute(0, 2, 1, 3)


def merge_published_ok(x_sa: torch.Tensor) -> bool:
    """bf16 (B, heads, N, d) view of dense token-major (B, N, heads, d) memory, N and heads * d multiples of 8"""
    if not (x_sa.is_cuda and x_sa.dtype == torch.bfloat16 and x_sa.dim() == 4):
        return False
    B, H, N, d = x_sa.shape
    return (N % 8 == 0 and (H * d) % 8 == 0 and x_sa.permute(0, 2, 1, 3).is_contiguous() and x_sa.data_ptr() % 16 == 0)


def merge_published(x_sa: torch.Tensor) -> torch.Tensor:
    """(B, heads, N, d) -> (B, N, C) as the published block merges its spatial branch (see _MergePublished)"""
    return _MergePublished.apply(x_sa)
