"""Autograd wrappers of the mesh-GNN edge kernels (include/py4cast_hip.h: p4c_edge_gather_add_fwd/bwd, p4c_segment_sum).

An InteractionNet layer (GraphLAM / HiLAM, config/CLI/model/graphlam.yaml:19-26) is
``m_e = MLP_e([e, x_s[src], x_r[dst]])``, ``agg = index_add(dst, m_e)``, ``x_r += MLP_n([x_r, agg])``.  torch runs the
edge side as index_select + cat + Linear and the aggregation as index_add_ (float atomics).  Here the first Linear of
MLP_e is distributed over the concat (nodes are projected once per NODE), the gather + add + activation is one kernel, and
the aggregation is a CSR segment sum (fixed order, bitwise reproducible).  No CPU fallback.
"""

from typing import Optional

import torch

from . import _lib as L

ACT = {None: 0, "none": 0, "relu": 1, "silu": 2}


class EdgeSet:
    """One edge list (src -> dst) with the receiver- and sender-sorted CSR both passes need.  Built once per graph."""

    def __init__(self, src: torch.Tensor, dst: torch.Tensor, n_src: int, n_dst: int):
        assert src.shape == dst.shape and src.dim() == 1
        self.n_src, self.n_dst, self.E = int(n_src), int(n_dst), int(src.numel())
        self.src = src.to(torch.int32).contiguous()
        self.dst = dst.to(torch.int32).contiguous()
        self.by_dst = _csr(self.dst, self.n_dst)
        self.by_src = _csr(self.src, self.n_src)

    def inv_degree(self, dtype: torch.dtype) -> torch.Tensor:
        """(n_dst, 1): 1 / number of incoming edges of every receiver (1 for a receiver without edges: its sum is zero anyway) --
        the factor that turns the segment sum into neural-lam's ``aggr="mean"`` (``mesh_aggr: mean``)"""
        key = ("inv_degree", dtype)
        cache = self.__dict__.setdefault("_derived", {})
        if key not in cache or cache[key].device != self.dst.device:
            off = self.by_dst[0].to(torch.int64)
            cache[key] = (1.0 / (off[1:] - off[:-1]).clamp_min(1).to(torch.float32)).to(dtype).unsqueeze(1)
        return cache[key]

    def to(self, device):
        self.__dict__.pop("_derived", None)
        for k in ("src", "dst"):
            setattr(self, k, getattr(self, k).to(device))
        self.by_dst = tuple(t.to(device) for t in self.by_dst)
        self.by_src = tuple(t.to(device) for t in self.by_src)
        return self


def _csr(index: torch.Tensor, n: int):
    idx = index.to(torch.int64)
    perm = torch.argsort(idx, stable=True).to(torch.int32)
    counts = torch.bincount(idx, minlength=n)
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=index.device)
    offsets[1:] = torch.cumsum(counts, 0)
    return offsets.to(torch.int32).contiguous(), perm.contiguous()


def _segment_sum_raw(msg: torch.Tensor, offsets: torch.Tensor, perm: Optional[torch.Tensor], n: int, out_dtype=None):
    L.require_cuda(msg, offsets)
    msg = msg.contiguous()
    E, C = msg.shape
    out_dtype = msg.dtype if out_dtype is None else out_dtype
    out = torch.empty(n, C, dtype=out_dtype, device=msg.device)
    L.call("p4c_segment_sum", L.ptr(msg), L.ptr(offsets), L.ptr(perm), None, L.ptr(out), n, E, C, L.dtype_code(msg.dtype),
           L.dtype_code(out_dtype), L.stream(msg.device),
           alg_bytes=E * C * msg.element_size() + n * C * out.element_size() + 4 * E * (perm is not None) + 4 * n)
    return out


def _segment_sum_pair_raw(msg: torch.Tensor, by_a, na: int, by_b, nb: int):
    """The sums of the same rows over two CSR structures (by_a = (offsets, perm), by_b likewise) in one launch."""
    L.require_cuda(msg)
    msg = msg.contiguous()
    E, C = msg.shape
    if E == 0 or na == 0 or nb == 0:
        return _segment_sum_raw(msg, *by_a, na), _segment_sum_raw(msg, *by_b, nb)
    out_a = torch.empty(na, C, dtype=msg.dtype, device=msg.device)
    out_b = torch.empty(nb, C, dtype=msg.dtype, device=msg.device)
    L.call("p4c_segment_sum_pair", L.ptr(msg), L.ptr(by_a[0]), L.ptr(by_a[1]), L.ptr(out_a), na, L.ptr(by_b[0]), L.ptr(by_b[1]),
           L.ptr(out_b), nb, E, C, L.dtype_code(msg.dtype), L.stream(msg.device),
           alg_bytes=2 * E * C * msg.element_size() + (na + nb) * C * msg.element_size() + 8 * E + 4 * (na + nb))
    return out_a, out_b


def _gather_raw(base, a, ia, b, ib, dh, act: int, E: int, C: int, like: torch.Tensor):
    out = torch.empty(E, C, dtype=like.dtype, device=like.device)
    name = "p4c_edge_gather_add_bwd" if dh is not None else "p4c_edge_gather_add_fwd"
    args = [L.ptr(base), L.ptr(a), L.ptr(ia), L.ptr(b), L.ptr(ib), L.ptr(out), E, C, L.dtype_code(like.dtype), act, L.stream(like.device)]
    if dh is not None:
        args = [L.ptr(dh)] + args
    row = C * like.element_size()
    nbytes = E * row * (1 + (base is not None) + (dh is not None))          # out, base, dh: one row per edge
    for t in (a, b):                                                          # gathered operands: every distinct row once
        if t is not None:
            nbytes += min(E, t.shape[0]) * row + 4 * E
    L.call(name, *args, alg_bytes=nbytes)
    return out


class _GatherAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, base, a, b, edges: EdgeSet, act: int):
        like = base if base is not None else (a if a is not None else b)
        L.require_cuda(like)
        tensors = [t.contiguous() if t is not None else None for t in (base, a, b)]
        for t in tensors:
            if t is not None and t.dtype != like.dtype:
                raise L.P4CError("edge_gather_add: base, a and b must share one dtype")
        base, a, b = tensors
        C = like.shape[-1]
        out = _gather_raw(base, a, edges.src if a is not None else None, b, edges.dst if b is not None else None, None, act,
                          edges.E, C, like)
        ctx.save_for_backward(*[t for t in tensors if t is not None])
        ctx.present = [t is not None for t in tensors]
        ctx.edges, ctx.act = edges, act
        return out

    @staticmethod
    def backward(ctx, dh):
        it = iter(ctx.saved_tensors)
        base, a, b = [next(it) if p else None for p in ctx.present]
        edges, act = ctx.edges, ctx.act
        dh = dh.contiguous()
        if act == 0:
            dpre = dh
        else:
            dpre = _gather_raw(base, a, edges.src if a is not None else None, b, edges.dst if b is not None else None, dh, act,
                               edges.E, dh.shape[-1], dh)
        dbase = dpre if (base is not None and ctx.needs_input_grad[0]) else None
        da = _segment_sum_raw(dpre, *edges.by_src, edges.n_src) if (a is not None and ctx.needs_input_grad[1]) else None
        db = _segment_sum_raw(dpre, *edges.by_dst, edges.n_dst) if (b is not None and ctx.needs_input_grad[2]) else None
        return dbase, da, db, None, None


def edge_gather_add(base: Optional[torch.Tensor], a: Optional[torch.Tensor], b: Optional[torch.Tensor], edges: EdgeSet,
                    act: Optional[str] = None) -> torch.Tensor:
    """``act(base[e] + a[src[e]] + b[dst[e]])`` -> (E, C).  base (E,C), a (n_src,C), b (n_dst,C); each may be None."""
    return _GatherAdd.apply(base, a, b, edges, ACT[act])


class _SegmentSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, msg, edges: EdgeSet):
        ctx.edges = edges
        return _segment_sum_raw(msg, *edges.by_dst, edges.n_dst)

    @staticmethod
    def backward(ctx, dout):
        edges = ctx.edges
        dout = dout.contiguous()
        return _gather_raw(None, None, None, dout, edges.dst, None, 0, edges.E, dout.shape[-1], dout), None


def aggregate_sum(msg: torch.Tensor, edges: EdgeSet) -> torch.Tensor:
    """``out[n] = sum_{e: dst[e] == n} msg[e]`` -> (n_dst, C)  (mesh_aggr: sum)."""
    return _SegmentSum.apply(msg, edges)
