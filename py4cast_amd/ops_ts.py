"""Autograd wrappers of the tall-skinny products (include/py4cast_hip.h: p4c_ts_gram / p4c_ts_apply) -- the two kernels the efficient
paired attention of UNETR++ is made of, forward and backward (each is the other's adjoint).  Token matrices are (B, heads, N, d)
VIEWS with unit stride in the last dimension and arbitrary strides elsewhere (slices of the qkvv projection, permuted views of
(B, N, C) tensors): they are addressed in place, nothing is made contiguous.  No CPU fallback."""

import torch

from . import _lib as L


def _strides(t: torch.Tensor):
    if t.dim() != 4 or (t.shape[-1] > 1 and t.stride(3) != 1):
        raise L.P4CError("tall-skinny ops take (B, heads, N, d) views with unit stride in the last dimension")
    return t.stride(0), t.stride(1), t.stride(2)


MAXD = 64   # columns per kernel call (csrc/tallskinny.hip); wider operands (decoder heads of 128 channels) go in column chunks


def _chunks(n):
    return [(i, min(i + MAXD, n)) for i in range(0, n, MAXD)]


def _gram_wide(x, y) -> bool:
    """Served by the matrix-core gram kernel (csrc/tallskinny.hip: gram_mfma_kernel): any width in one launch."""
    if not L.lib().p4c_ts_gram_wide_ok(L.dtype_code(x.dtype), L.dtype_code(y.dtype), x.shape[-1], y.shape[-1]):
        return False
    al = lambda t: all(v % 8 == 0 for v in _strides(t)) and t.data_ptr() % 16 == 0
    return al(x) and al(y)


def _gram_call(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    B, H, N, d = x.shape
    e = y.shape[-1]
    ns = L.lib().p4c_ts_gram_splits(N)
    part = torch.empty(B, ns, H, d, e, dtype=torch.float32, device=x.device)
    xs, ys = _strides(x), _strides(y)
    L.call("p4c_ts_gram", L.ptr(x), L.dtype_code(x.dtype), *xs, L.ptr(y), L.dtype_code(y.dtype), *ys, L.ptr(part), B, H, N, d, e,
           L.stream(x.device), alg_bytes=B * H * N * (d * x.element_size() + e * y.element_size()))
    return part.sum(dim=1) if ns > 1 else part[:, 0]


def _gram_raw(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    d, e = x.shape[-1], y.shape[-1]
    if (d <= MAXD and e <= MAXD) or _gram_wide(x, y):
        return _gram_call(x, y)
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


def _apply_call(x, m, ov, B, H, N, d, e, accumulate):
    """out (+)= x @ m on the matrix-core kernel; a transposed view of a stored matrix is read as stored (p4c_ts_apply_mt), anything
    else is made contiguous first"""
    args_x = (L.ptr(x), L.dtype_code(x.dtype), *_strides(x))
    args_o = (L.ptr(ov), L.dtype_code(ov.dtype), *_strides(ov))
    nbytes = B * H * N * (d * x.element_size() + e * ov.element_size() * (1 + int(accumulate)))
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
        tot = part.sum(dim=1) if ns > 1 else part[:, 0]
        ctx.save_for_backward(x, y)
        return tot[..., : d * e].reshape(B, H, d, e), tot[..., d * e : d * e + d], tot[..., d * e + d :]

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
    (bf16 flavour, d, p <= 64): qkvv (B, N, 4, heads, d) -> (x_sa, x_ca), both (B, heads, N, d) token-major.  The pieces are the
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
        # q^T k with the squared column norms of q and k
        ns = L.lib().p4c_ts_gram_splits(N)
        part = torch.empty(B, ns, H, d * d + 2 * d, dtype=torch.float32, device=qkvv.device)
        L.call("p4c_ts_gram_norms", L.ptr(q), L.dtype_code(dt), *_strides(q), L.ptr(k), L.dtype_code(dt), *_strides(k), L.ptr(part), B, H, N, d, d,
               L.stream(qkvv.device), alg_bytes=B * H * N * 2 * d * q.element_size())
        tot = part.sum(dim=1) if ns > 1 else part[:, 0]
        G = tot[..., : d * d].reshape(B, H, d, d).contiguous()
        nq2, nk2 = tot[..., d * d: d * d + d].contiguous(), tot[..., d * d + d:].contiguous()   # (B,H,d): |q columns|^2, |k columns|^2
        # token-axis projection of k and v_sa (shared weights E = F): KP[b,h] = k[b,h]^T W^T + bias, (d x p) per head
        from .ops_rows import weight_as

        W16 = weight_as(W, dt)                                               # (p, N)
        if _token_proj_native(qkvv, p):
            # round 6 (diagnostic route, measured no faster): the projection IS a gram product of the head's token matrix with the
            # (N x p) matrix W^T shared by all heads (strides 0 over sample and head): k and v_sa are read IN PLACE inside qkvv by the
            # tall-skinny kernel -- no (2, B, N, C) gather of them (16 MB per block at the first stage)
            Wt = weight_as(W, dt, transposed=True)                           # (N, p), once per parameter version
            Wv = Wt.view(1, 1, N, p).expand(B, H, N, p)
            proj = torch.empty(2, B, H, d, p, dtype=torch.float32, device=qkvv.device)
            for i, x in enumerate((k, vsa)):
                part = _gram_partial(x, Wv)                                  # (B, splits, H, d, p)
                if part.shape[1] > 1:
                    torch.sum(part, dim=1, out=proj[i])
                else:
                    proj[i].copy_(part[:, 0])
            proj += bias.float()
            kv = None
        else:
            kv = qkvv[:, :, 1::2].permute(2, 0, 1, 3, 4).reshape(2, B, N, C)      # k and v_sa token-major, ONE strided copy: (2,B,N,C)
            proj = (kv.transpose(-1, -2) @ W16.t()).float() + bias.float()       # (2,B,C,p): KP and VP are its two contiguous halves
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
        ctx.native_proj = kv is None
        ctx.save_for_backward(qkvv, W16, Wt if kv is None else kv, G, nq2, nk2, KP, VP, t1f, t2f, At, Mq, nrm, S)
        ctx.meta = (W.dtype, bias.dtype, t1.shape, t1.dtype)
        return x_sa, x_ca

    @staticmethod
    def backward(ctx, dx_sa, dx_ca):
        qkvv, W16, kv, G, nq2, nk2, KP, VP, t1f, t2f, At, Mq, nrm, S = ctx.saved_tensors
        wdt, bdt, tshape, tdt = ctx.meta
        B, N, _, H, d = qkvv.shape
        C, p, dt = H * d, KP.shape[-1], qkvv.dtype
        q, k, vca, vsa = (qkvv[:, :, i].permute(0, 2, 1, 3) for i in range(4))
        ok = lambda t: t.stride(3) == 1 and all(v % 8 == 0 for v in _strides(t)) and t.data_ptr() % 16 == 0   # noqa: E731
        dx_sa = dx_sa if ok(dx_sa) else dx_sa.contiguous()
        dx_ca = dx_ca if ok(dx_ca) else dx_ca.contiguous()
        dqkvv = torch.empty_like(qkvv)
        dq, dk, dvca, dvsa = (dqkvv[:, :, i].permute(0, 2, 1, 3) for i in range(4))
        # channel branch x_ca = v_ca At
        _apply_into(dvca, dx_ca, At.transpose(-1, -2), False)
        dAt = _gram_raw(vca, dx_ca)
        # spatial branch x_sa = softmax(q Mq) VP^T
        dL = _apply_softmax(dx_sa, VP, 2, S)                                 # dS = dx VP, softmax adjoint in the epilogue
        dVP = _gram_raw(dx_sa, S)                                            # d(VP^T) = S^T dx, i.e. dVP = dx^T S (B,H,d,p)
        _apply_into(dq, dL, Mq.transpose(-1, -2), False)                     # dq, first contribution
        dMq = _gram_raw(q, dL)
        # small matrices
        dG = torch.empty(B, H, d, d, dtype=torch.float32, device=G.device)
        dn = torch.empty(2, B, H, d, dtype=torch.float32, device=G.device)
        dKP = torch.empty_like(KP)
        dtp = torch.empty(2, B, H, dtype=torch.float32, device=G.device)
        L.call("p4c_epa_small_bwd", L.ptr(G), L.ptr(nq2), L.ptr(nk2), L.ptr(KP), L.ptr(t1f), L.ptr(t2f), L.ptr(At), L.ptr(nrm[0]), L.ptr(nrm[1]),
               L.ptr(dAt.float().contiguous()), L.ptr(dMq.float().contiguous()), L.ptr(dG), L.ptr(dn[0]), L.ptr(dn[1]), L.ptr(dKP), L.ptr(dtp[0]),
               L.ptr(dtp[1]), B, H, d, p, 1, L.stream(G.device))
        dts = dtp.sum(dim=1)
        # token-axis projection: proj = kv^T W16^T + bias
        g = torch.stack([dKP.reshape(B, C, p), dVP.reshape(B, C, p)], dim=0)         # (2,B,C,p) fp32
        dbias = g.sum(dim=(0, 1, 2)).to(bdt)
        if ctx.native_proj:
            # the adjoints as tall-skinny products on the operands in place: dk = W^T dKP^T and dv_sa = W^T dVP^T per head (apply with
            # the shared (N x p) matrix as the token operand: first contributions, written straight into dqkvv), dW^T = sum over
            # samples of k[b] (N x C) dKP[b] (C x p) + the same for v_sa (apply over ALL heads' channels at once, accumulated in a
            # fixed order into one fp32 (N x p) matrix) -- no bf16 copy of g, no (2,B,N,C) gradient tensor, no scatter copy
            Wt = kv
            Wv = Wt.view(1, 1, N, p).expand(B, H, N, p)
            _apply_into(dk, Wv, dKP.transpose(-1, -2), False)
            _apply_into(dvsa, Wv, dVP.transpose(-1, -2), False)
            dWt = torch.empty(1, N, 1, p, dtype=torch.float32, device=G.device)
            dWv = dWt.permute(0, 2, 1, 3)                                            # (1,1,N,p) token-major view
            first = True
            for b in range(B):
                for i, gm in ((1, dKP), (3, dVP)):
                    xb = qkvv[b:b + 1, :, i].reshape(1, N, 1, C).permute(0, 2, 1, 3)   # (1,1,N,C): all heads' channels of sample b, in place
                    _apply_into(dWv, xb, gm[b].reshape(1, 1, C, p), not first)
                    first = False
            dW = dWt.view(N, p).t().to(wdt)                                          # (p, N)
        else:
            g16 = g.to(dt)
            dkv = (g16 @ W16).transpose(-1, -2)                                          # (2,B,N,C) view of (2,B,C,N)
            dW = torch.bmm(g16.reshape(2 * B, C, p).transpose(1, 2), kv.reshape(2 * B, N, C).transpose(1, 2)).sum(dim=0).to(wdt)   # (p,N)
            dqkvv[:, :, 1::2].copy_(dkv.reshape(2, B, N, H, d).permute(1, 2, 0, 3, 4))    # dk (first contribution) and dv_sa: one copy
        # q^T k and the norms: dq += k dG^T + 2 q diag(dnq2),  dk += q dG + 2 k diag(dnk2)
        D = dn.unsqueeze(-1) * _two_eye(d, dn.device)                                # (2,B,H,d,d) = 2 diag(dn), one launch
        _apply_into(dq, k, dG.transpose(-1, -2), True)
        _apply_into(dq, q, D[0], True)
        _apply_into(dk, q, dG, True)
        _apply_into(dk, k, D[1], True)
        return dqkvv, dW, dbias, dts[0].view(tshape).to(tdt), dts[1].view(tshape).to(tdt)


def _gram_partial(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """the per-split partials (B, splits, H, d, e) of X^T Y (p4c_ts_gram): the caller sums them where it wants the result"""
    B, H, N, d = x.shape
    e = y.shape[-1]
    ns = L.lib().p4c_ts_gram_splits(N)
    part = torch.empty(B, ns, H, d, e, dtype=torch.float32, device=x.device)
    L.call("p4c_ts_gram", L.ptr(x), L.dtype_code(x.dtype), *_strides(x), L.ptr(y), L.dtype_code(y.dtype), *_strides(y), L.ptr(part), B, H, N, d, e,
           L.stream(x.device), alg_bytes=B * H * N * d * x.element_size() + N * e * y.element_size())
    return part


TOKEN_PROJ_MIN_TOKENS = 2048   # from here on the (2, B, N, C) gather of k / v_sa is worth avoiding (below: the library GEMM on a small copy)


def _token_proj_native(qkvv: torch.Tensor, p: int) -> bool:
    """EPA's token-axis projection on the tall-skinny kernels, k / v_sa read in place (see _EpaCore.forward): the stages with many
    tokens, whose whole token rows (C = heads x d channels) fit the matrix-core apply kernel (C <= 256) for the weight gradient"""
    B, N, _, H, d = qkvv.shape
    C = H * d
    # Measured in round 6 (profiles/r06_ab_runs.txt 6): the UNETR++ step is NOT faster this way -- 137.8 ms against 136.8 with the
    # gather + library GEMM (the shared (N x p) operand is re-read by every (sample, head) group, the weight gradient is four
    # accumulating passes over an fp32 (N x p) matrix) -- so the library route stays the product's; this one is a diagnostic switch.
    if L.diag_switch("P4C_EPA_NATIVE_PROJ") != "1" or N < TOKEN_PROJ_MIN_TOKENS or C > 256 or C % 8 or p % 8:
        return False
    lib, bf, f32 = L.lib(), L.dtype_code(torch.bfloat16), L.dtype_code(torch.float32)
    return bool(lib.p4c_ts_gram_wide_ok(bf, bf, d, p) and lib.p4c_ts_apply_wide_ok(bf, bf, p, d) and lib.p4c_ts_apply_wide_ok(bf, f32, C, p))


def epa_core_ok(qkvv: torch.Tensor, p: int) -> bool:
    """bf16 qkvv (B, N, 4, heads, d) with d, p multiples of 8 up to 64 and 16-byte aligned head rows."""
    if qkvv.dim() != 5 or qkvv.dtype != torch.bfloat16 or not qkvv.is_contiguous():
        return False
    d = qkvv.shape[-1]
    return d % 8 == 0 and p % 8 == 0 and d <= 64 and p <= 64 and 64 % (d // 8) == 0 and spatial_fused_ok(qkvv[:, :, 0].permute(0, 2, 1, 3), p)


def epa_core(qkvv, W, bias, t1, t2):
    """(x_sa, x_ca) of an EPA block from its qkvv projection, the token-axis Linear E (weight (p, N), bias) and the temperatures."""
    return _EpaCore.apply(qkvv, W, bias, t1, t2)
