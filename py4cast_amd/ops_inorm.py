"""InstanceNorm2d(affine) + LeakyReLU (+ residual) on features-last tensors as ONE autograd node on the kernels of csrc/inorm.hip
(include/py4cast_hip.h: p4c_inorm_reduce / p4c_inorm_apply): two streaming passes each way instead of torch's reductions and
elementwise chains (the UNETR decoder of SwinUNetR spent a third of its step in them).  No CPU fallback."""

from typing import Optional

import torch

from . import _lib as L


_TICKETS = {}      # device -> (zeroed uint32 slots, next slot): the tickets of the in-kernel finalize (csrc/inorm.hip: InFin)
# Measured in round 6 (profiles/r06_ab_runs.txt 7) and NOT taken: SwinUNETR 26.8 ms with the in-kernel finalize against 25.65 with the
# finalize launch, UNETR++ 143.9 against 137.7 -- the standalone finalize spreads its (sample, channel chunk) pairs over workgroups,
# the last workgroup of the reduce launch walks them one after the other (C = 384: 12 serial slice sums) behind write-through stores and
# a fence, and under HIP-graph replay the small dependent launch it saves costs ~2 us.  False = the product route; True is kept for the
# parity tests of the entry points (tests/test_inorm_fin_gpu.py).
FUSED_FINALIZE = False


def next_ticket(device) -> int:
    """address of a zeroed uint32 for one launch with an in-kernel finalize.  The last workgroup of that launch resets its ticket, so
    a slot is reusable as soon as the launch has finished; launches of one stream run in order and far fewer than the 1 024 slots are
    ever in flight, so the slots are dealt round-robin.  Allocated once per device OUTSIDE any HIP-graph capture (a captured launch
    keeps the address)."""
    key = str(device)
    ent = _TICKETS.get(key)
    if ent is None:
        if torch.cuda.is_current_stream_capturing():
            raise L.P4CError("ops_inorm: the ticket pool must exist before a HIP-graph capture (run one eager step first)")
        ent = _TICKETS[key] = [torch.zeros(1024, dtype=torch.int32, device=device), 0]
    ent[1] = (ent[1] + 1) % 1024
    return ent[0].data_ptr() + 4 * ent[1]


def _partials(x, dy, y, mean, rstd, slope):
    """p4c_inorm_reduce: per-block partial sums (B, nb, 2, C)"""
    B, C = x.shape[0], x.shape[-1]
    N = x.numel() // (B * C)
    nb = L.lib().p4c_inorm_blocks(N, C)
    part = torch.empty(B, nb, 2, C, dtype=torch.float32, device=x.device)
    L.call("p4c_inorm_reduce", L.ptr(x), L.ptr(dy), L.ptr(y), L.ptr(mean), L.ptr(rstd), float(slope), L.ptr(part), L.dtype_code(x.dtype),
           B, N, C, L.stream(x.device))
    return part, nb, N


def _reduce(x, dy, y, mean, rstd, slope):
    part, _, N = _partials(x, dy, y, mean, rstd, slope)
    return part.sum(dim=1), N       # (B, 2, C): fixed order


def _f32(p):
    return p.detach() if p.dtype == torch.float32 else p.detach().float()


def _finalize_fwd(part, nb, N, C, groups, weight, bias, eps):
    """mean, rstd, scale, shift (B, C) from the partial sums in ONE launch (p4c_inorm_finalize_fwd)"""
    B = part.shape[0]
    st = torch.empty(4, B, C, dtype=torch.float32, device=part.device)
    L.call("p4c_inorm_finalize_fwd", L.ptr(part), nb, B, N, C, groups, L.ptr(_f32(weight)), L.ptr(_f32(bias)), float(eps), L.ptr(st[0]), L.ptr(st[1]),
           L.ptr(st[2]), L.ptr(st[3]), L.stream(part.device))
    return st[0], st[1], st[2], st[3]


def _finalize_bwd(part, nb, N, C, groups, weight, rstd):
    """c1, c2 (B, C) and dgamma, dbeta (C) from the backward partial sums in ONE launch (p4c_inorm_finalize_bwd)"""
    B = part.shape[0]
    co = torch.empty(2, B, C, dtype=torch.float32, device=part.device)
    dgb = torch.empty(2, C, dtype=torch.float32, device=part.device)
    L.call("p4c_inorm_finalize_bwd", L.ptr(part), nb, B, N, C, groups, L.ptr(_f32(weight)) if groups else None, L.ptr(rstd) if groups else None,
           L.ptr(co[0]), L.ptr(co[1]), L.ptr(dgb[0]), L.ptr(dgb[1]), L.stream(part.device))
    return co[0], co[1], dgb[0], dgb[1]


class _InstNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, res, eps, slope):
        L.require_cuda(x)
        x = x.contiguous()
        res_c = None if res is None else res.contiguous()
        B, C = x.shape[0], x.shape[-1]
        if FUSED_FINALIZE:      # statistics finished by the reduce launch's last workgroup (no finalize launch)
            N = x.numel() // (B * C)
            nb = L.lib().p4c_inorm_blocks(N, C)
            part = torch.empty(B, nb, 2, C, dtype=torch.float32, device=x.device)
            st = torch.empty(4, B, C, dtype=torch.float32, device=x.device)
            L.call("p4c_inorm_reduce_finalize_fwd", L.ptr(x), L.ptr(part), next_ticket(x.device), L.ptr(_f32(weight)), L.ptr(_f32(bias)), float(eps),
                   L.ptr(st[0]), L.ptr(st[1]), L.ptr(st[2]), L.ptr(st[3]), L.dtype_code(x.dtype), B, N, C, L.stream(x.device),
                   alg_bytes=x.numel() * x.element_size())
            mean, rstd, scale, shift = st[0], st[1], st[2], st[3]
        else:
            part, nb, N = _partials(x, None, None, None, None, slope)
            mean, rstd, scale, shift = _finalize_fwd(part, nb, N, C, 0, weight, bias, eps)
        y = torch.empty_like(x)
        L.call("p4c_inorm_apply", L.ptr(x), L.ptr(res_c), None, None, L.ptr(scale), L.ptr(shift), None, None, None, None, float(slope),
               L.ptr(y), None, L.dtype_code(x.dtype), B, N, C, L.stream(x.device))
        ctx.save_for_backward(x, y, mean, rstd, scale)
        ctx.slope, ctx.has_res, ctx.pdtype = slope, res is not None, weight.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, rstd, scale = ctx.saved_tensors
        B, C = x.shape[0], x.shape[-1]
        dy = dy.contiguous()
        if FUSED_FINALIZE:
            N = x.numel() // (B * C)
            nb = L.lib().p4c_inorm_blocks(N, C)
            part = torch.empty(B, nb, 2, C, dtype=torch.float32, device=x.device)
            co = torch.empty(2, B, C, dtype=torch.float32, device=x.device)
            dgb = torch.empty(2, C, dtype=torch.float32, device=x.device)
            L.call("p4c_inorm_reduce_finalize_bwd", L.ptr(x), L.ptr(dy), L.ptr(y), L.ptr(mean), L.ptr(rstd), float(ctx.slope), L.ptr(part),
                   next_ticket(x.device), L.ptr(co[0]), L.ptr(co[1]), L.ptr(dgb[0]), L.ptr(dgb[1]), L.dtype_code(x.dtype), B, N, C,
                   L.stream(x.device), alg_bytes=3 * x.numel() * x.element_size())
            m1, m2, dgamma, dbeta = co[0], co[1], dgb[0], dgb[1]
        else:
            part, nb, N = _partials(x, dy, y, mean, rstd, ctx.slope)
            m1, m2, dgamma, dbeta = _finalize_bwd(part, nb, N, C, 0, None, None)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        L.call("p4c_inorm_apply", L.ptr(x), None, L.ptr(dy), L.ptr(y), L.ptr(scale), None, L.ptr(mean), L.ptr(rstd), L.ptr(m1), L.ptr(m2),
               float(ctx.slope), L.ptr(dx), L.ptr(dres), L.dtype_code(x.dtype), B, N, C, L.stream(x.device))
        return dx, dgamma.to(ctx.pdtype), dbeta.to(ctx.pdtype), dres, None, None


def supported(x: torch.Tensor) -> bool:
    return x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.shape[-1] % 4 == 0 and x.shape[-1] <= 1024


def instance_norm_act(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5, slope: float = 0.01,
                      res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``leaky_relu(instance_norm(x) * weight + bias (+ res), slope)`` for x (B, *spatial, C) features-last; slope = 1 gives the plain
    affine instance norm."""
    return _InstNormAct.apply(x, weight, bias, res, float(eps), float(slope))


class _GroupNorm(torch.autograd.Function):
    """GroupNorm(groups, C)(x) * weight + bias on a features-last tensor: per-(sample, channel) sums from p4c_inorm_reduce, the group
    statistics from them (a few hundred numbers, torch), one streaming apply pass each way (p4c_inorm_apply).  With
    xhat = (x - mean_g) rstd_g, M1 = mean_g(gamma dy), M2 = mean_g(gamma dy xhat):  dx = rstd_g (gamma_c dy - M1 - xhat M2)."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps):
        L.require_cuda(x)
        x = x.contiguous()
        B, C = x.shape[0], x.shape[-1]
        ctx.native = C // groups <= 64
        if ctx.native:   # statistics of the groups in one launch (p4c_inorm_finalize_fwd)
            part, nb, N = _partials(x, None, None, None, None, 1.0)
            mean, rstd, scale, shift = _finalize_fwd(part, nb, N, C, groups, weight, bias, eps)
        else:
            sums, N = _reduce(x, None, None, None, None, 1.0)                   # (B, 2, C): sum x, sum x^2
            n = float(N * (C // groups))
            gs = sums.view(B, 2, groups, C // groups).sum(dim=-1)               # (B, 2, G)
            mean_g = gs[:, 0] / n
            rstd_g = torch.rsqrt((gs[:, 1] / n - mean_g * mean_g).clamp_min(0) + eps)
            mean = mean_g.repeat_interleave(C // groups, dim=1).contiguous()    # (B, C)
            rstd = rstd_g.repeat_interleave(C // groups, dim=1).contiguous()
            scale = (rstd * weight.float()).contiguous()
            shift = (bias.float() - mean * scale).contiguous()
        y = torch.empty_like(x)
        L.call("p4c_inorm_apply", L.ptr(x), None, None, None, L.ptr(scale), L.ptr(shift), None, None, None, None, 1.0, L.ptr(y), None,
               L.dtype_code(x.dtype), B, N, C, L.stream(x.device))
        ctx.save_for_backward(x, mean, rstd, scale, weight)
        ctx.groups, ctx.pdtype = groups, weight.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, scale, weight = ctx.saved_tensors
        B, C, G = x.shape[0], x.shape[-1], ctx.groups
        dy = dy.contiguous()
        if ctx.native:
            part, nb, N = _partials(x, dy, x, mean, rstd, 1.0)                   # sum dy, sum dy xhat  (slope 1: y is not looked at)
            c1, c2, dgamma, dbeta = _finalize_bwd(part, nb, N, C, G, weight, rstd)
        else:
            sums, N = _reduce(x, dy, x, mean, rstd, 1.0)                        # (B, 2, C)
            n = float(N * (C // G))
            gam = weight.float()
            m = (sums * gam).view(B, 2, G, C // G).sum(dim=-1) / n              # (B, 2, G): M1, M2
            M1 = m[:, 0].repeat_interleave(C // G, dim=1)
            M2 = m[:, 1].repeat_interleave(C // G, dim=1)
            c1 = (rstd * M1).contiguous()                                       # dx = scale dy - c1 - (x - mean) c2
            c2 = (rstd * rstd * M2).contiguous()
            dgamma, dbeta = sums[:, 1].sum(dim=0), sums[:, 0].sum(dim=0)
        dx = torch.empty_like(x)
        L.call("p4c_inorm_apply", L.ptr(x), None, L.ptr(dy), L.ptr(x), L.ptr(scale), None, L.ptr(mean), None, L.ptr(c1), L.ptr(c2), 1.0,
               L.ptr(dx), None, L.dtype_code(x.dtype), B, N, C, L.stream(x.device))
        return dx, dgamma.to(ctx.pdtype), dbeta.to(ctx.pdtype), None, None


def group_norm(x: torch.Tensor, groups: int, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """``F.group_norm`` for x (B, *spatial, C) features-last (statistics over the spatial positions and the C / groups channels of
    a group, per sample)."""
    if x.shape[-1] % groups:
        raise L.P4CError(f"group_norm: {x.shape[-1]} channels do not divide into {groups} groups")
    return _GroupNorm.apply(x, weight, bias, int(groups), float(eps))
