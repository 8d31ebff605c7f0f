"""
The "flat" fused AR-step kernels (csrc/losses.hip, round 4): state update + border forcing + weighted loss (+ next network input +
saved loss gradients) and its backward for ANY feature count -- the reference's shipped Titan configuration has F = 21
(config/CLI/dataset/titan.yaml:38-76), which the 16-byte kernels (F % 4 == 0) do not serve.  Same arithmetic per element as the
kernels they stand in for, so:
  * F % 4 == 0, P4C_FORCE_FLAT_STEP=1: new state, next input, saved loss gradients, dy and dprev equal the 16-byte kernels' BIT FOR BIT;
  * F % 4 != 0: new state / dy / dprev equal the scalar kernels' (P4C_NO_FLAT_STEP=1) bit for bit, the next input equals p4c_build_x
    on the new state, the saved loss gradients equal the element formula;
the loss is the same sum in another order (<= 1e-6).
"""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("diag_library")]   # (flips P4C_* A/B switches: diagnostic build)


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-30))


def _case(dev, B, N, F, Fs, Ff, cpad, dt, seed=29):
    g = torch.Generator(device=dev).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    ru = lambda *s: torch.rand(*s, generator=g, device=dev)
    c = dict(y=rn(B, N, 64).to(dt), prev=rn(B, N, F), tgt=rn(B, N, F), std=ru(F) + 0.5, mean=rn(F) * 0.01,
             interior=(ru(N) > 0.2).float(), weights=ru(F) + 0.5, statics=ru(B, N, max(Fs, 1)), forcing=ru(B, N, max(Ff, 1)),
             g1=rn(B, N, F), g2=rn(B, N, 64).to(dt), gloss=ru(B) + 0.5)
    c["border"] = 1.0 - c["interior"]
    return c


def _fwd(L, c, B, N, F, Fs, Ff, cpad, dt, kind, border, scaled, nxt, lg):
    dev = c["y"].device
    code = L.dtype_code(dt)
    ws = torch.empty(L.lib().p4c_loss_workspace_bytes(B, 1, N, 1) // 4, dtype=torch.float32, device=dev)
    ns, loss = torch.empty(B, N, F, device=dev), torch.empty(B, device=dev)
    xn = torch.full((B, N, cpad), 7.0, device=dev).to(dt) if nxt else None
    lgr = torch.full((B, N, F), 7.0, device=dev).bfloat16() if lg else None
    std, mean = (c["std"], c["mean"]) if scaled else (None, None)
    args = [L.ptr(c["prev"]), N * F, L.ptr(c["y"]), code, 64, L.ptr(c["tgt"]), N * F, L.ptr(std), L.ptr(mean),
            L.ptr(c["border"] if border else None), L.ptr(c["interior"]), L.ptr(ns), N * F, L.ptr(c["weights"]), float(c["interior"].sum()),
            None, kind, L.MASK_NONE, L.ptr(loss), 1, L.ptr(ws), B, N, F, 1.0]
    st = L.stream(dev)
    if lg:
        L.call("p4c_ar_update_loss_fwd_next_saved", *args, L.ptr(xn), cpad, L.ptr(c["statics"]), N * max(Fs, 1), Fs, L.ptr(c["forcing"]),
               N * max(Ff, 1), Ff, L.ptr(lgr), N * F, st)
    elif nxt:
        L.call("p4c_ar_update_loss_fwd_next", *args, L.ptr(xn), cpad, L.ptr(c["statics"]), N * max(Fs, 1), Fs, L.ptr(c["forcing"]),
               N * max(Ff, 1), Ff, st)
    else:
        L.call("p4c_ar_update_loss_fwd", *args, st)
    torch.cuda.synchronize()
    return ns, loss, xn, lgr


def _bwd(L, c, B, N, F, dt, kind, force_border, scaled, first, last, lgr=None, ns=None):
    dev = c["y"].device
    code = L.dtype_code(dt)
    dy = torch.full((B, N, 64), 7.0, device=dev).to(dt)
    dprev = None if first else torch.full((B, N, F), 7.0, device=dev)
    g1 = None if last else c["g1"]
    g2 = None if last else c["g2"]
    std = c["std"] if scaled else None
    st = L.stream(dev)
    if lgr is not None:
        L.call("p4c_ar_update_loss_bwd_saved", L.ptr(g1), N * F, L.ptr(g2), code, 64, L.ptr(c["gloss"]), 1, L.ptr(lgr), N * F, L.ptr(std),
               L.ptr(c["interior"]), int(force_border), L.ptr(c["weights"]), float(c["interior"].sum()), None, kind, L.MASK_NONE, L.ptr(dy),
               code, 64, L.ptr(dprev), N * F, B, N, F, 1.0, st)
    else:
        L.call("p4c_ar_update_loss_bwd", L.ptr(g1), N * F, L.ptr(g2), code, 64, L.ptr(c["gloss"]), 1, L.ptr(ns), N * F, L.ptr(c["tgt"]), N * F,
               L.ptr(std), L.ptr(c["interior"]), int(force_border), L.ptr(c["weights"]), float(c["interior"].sum()), None, kind, L.MASK_NONE,
               L.ptr(dy), code, 64, L.ptr(dprev), N * F, B, N, F, 1.0, st)
    torch.cuda.synchronize()
    return dy, dprev


def _bits(t):
    return t.view(torch.int16) if t.dtype == torch.bfloat16 else t


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,N,Fs,Ff,cpad,kind,border,scaled,nxt,lg", [
    (2, 64 * 96, 4, 5, 96, 0, True, True, True, True),
    (1, 1000, 4, 5, 96, 1, True, True, True, True),        # a partial last tile (1000 = 15 x 64 + 40)
    (2, 2048, 4, 5, 96, 0, False, True, False, True),      # no border forcing, last AR step
    (3, 772, 4, 8, 96, 0, True, False, True, False),       # unscaled update, no saved gradients
])
def test_flat_kernels_equal_the_16_byte_kernels(gpu_device, monkeypatch, dt, B, N, Fs, Ff, cpad, kind, border, scaled, nxt, lg):
    """F = 60: the flat kernels forced on (P4C_FORCE_FLAT_STEP=1) against the 16-byte kernels, forward and both backward forms."""
    from py4cast_amd import _lib as L

    F = 60
    c = _case(gpu_device, B, N, F, Fs, Ff, cpad, dt)
    res = {}
    for force in ("0", "1"):
        monkeypatch.setenv("P4C_FORCE_FLAT_STEP", force)
        ns, loss, xn, lgr = _fwd(L, c, B, N, F, Fs, Ff, cpad, dt, kind, border, scaled, nxt, lg)
        dy, dprev = _bwd(L, c, B, N, F, dt, kind, border, scaled, first=False, last=False, ns=ns)
        dy_l, _ = _bwd(L, c, B, N, F, dt, kind, border, scaled, first=True, last=True, ns=ns)
        dys, dps = (_bwd(L, c, B, N, F, dt, kind, border, scaled, first=False, last=False, lgr=lgr) if lg else (None, None))
        res[force] = (ns, loss, xn, lgr, dy, dprev, dy_l, dys, dps)
    a, b = res["0"], res["1"]
    assert torch.equal(a[0], b[0])
    assert rel_err(b[1], a[1]) < 1e-6
    for i in (2, 3, 4, 6, 7):
        if a[i] is not None:
            assert torch.equal(_bits(a[i]), _bits(b[i])), i
    for i in (5, 8):
        if a[i] is not None:
            assert torch.equal(a[i], b[i]), i


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,N,F,Fs,Ff,cpad,kind", [
    (2, 64 * 80, 21, 4, 21, 64, 0),      # the shipped Titan feature counts (C_in = 46 -> 64)
    (1, 1000, 21, 4, 21, 64, 1),         # partial last tile, L1
    (2, 2052, 5, 3, 2, 32, 0),           # odd static / forcing counts
    (2, 772, 62, 1, 1, 64, 0),           # rows nearly full
    (3, 64, 13, 0, 0, 32, 0),            # one tile, nothing but the state in the next input
])
def test_flat_kernels_for_feature_counts_off_the_16_byte_grid(gpu_device, monkeypatch, dt, B, N, F, Fs, Ff, cpad, kind):
    """F % 4 != 0: flat kernels (default) against the scalar kernels (P4C_NO_FLAT_STEP=1) for what those can do (the plain fused step
    and its backward); the next input against p4c_build_x on the new state; the saved loss gradients against the element formula, and
    the backward from them against the backward that recomputes them from the SAME bf16-rounded values."""
    from py4cast_amd import _lib as L
    from py4cast_amd import ops

    c = _case(gpu_device, B, N, F, Fs, Ff, cpad, dt)
    monkeypatch.setenv("P4C_NO_FLAT_STEP", "1")
    ns0, loss0, _, _ = _fwd(L, c, B, N, F, Fs, Ff, cpad, dt, kind, True, True, False, False)
    dy0, dp0 = _bwd(L, c, B, N, F, dt, kind, True, True, first=False, last=False, ns=ns0)
    monkeypatch.setenv("P4C_NO_FLAT_STEP", "0")
    ns1, loss1, xn, lgr = _fwd(L, c, B, N, F, Fs, Ff, cpad, dt, kind, True, True, True, True)
    dy1, dp1 = _bwd(L, c, B, N, F, dt, kind, True, True, first=False, last=False, ns=ns1)
    assert torch.equal(ns1, ns0) and rel_err(loss1, loss0) < 1e-6
    assert torch.equal(_bits(dy1), _bits(dy0)) and torch.equal(dp1, dp0)
    # next input = build_x(new state | statics | forcing | zero padding) in the row dtype
    if Fs > 0 and Ff > 0:
        want = ops.build_x(ns1.view(B, 1, N, F), c["statics"][..., :Fs].contiguous(), c["forcing"][..., :Ff].contiguous(), c_pad=cpad, dtype=dt)
    else:
        want = torch.zeros(B, N, cpad, device=gpu_device)
        want[..., :F] = ns1
        want = want.to(dt)
    assert torch.equal(_bits(xn), _bits(want.reshape(B, N, cpad)))
    # saved loss gradients: d loss_elem / d pred, rounded to bf16
    d = ns1 - c["tgt"]
    lg_want = (2.0 * d if kind == 0 else torch.sign(d)).bfloat16()
    assert torch.equal(_bits(lgr), _bits(lg_want))
    # backward from the saved gradients: same formula with the bf16 value in place of the recomputed one
    dys, dps = _bwd(L, c, B, N, F, dt, kind, True, True, first=False, last=False, lgr=lgr)
    im = c["interior"][None, :, None]
    g = (c["gloss"] / float(c["interior"].sum()))[:, None, None] * im * c["weights"] * lgr.float() + c["g1"] + c["g2"][..., :F].float()
    gp = g * im
    assert rel_err(dps, gp) < 1e-6
    assert rel_err(dys[..., :F].float(), (gp * c["std"]).to(dt).float()) < 1e-6
    assert float(dys[..., F:].float().abs().max()) == 0.0        # channels >= F of dy are zero


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,N,T_in,F,Fs,Ff,cpad", [
    (2, 64 * 80, 1, 21, 4, 21, 64),      # the shipped Titan feature counts (C_in = 46 -> 64)
    (1, 1000, 1, 21, 4, 21, 64),         # a partial last tile
    (2, 2052, 2, 5, 3, 2, 32),           # two input time steps, odd static / forcing counts
    (3, 772, 1, 62, 1, 1, 64),           # rows full
    (2, 512 * 640, 1, 21, 4, 21, 64),    # the Titan grid
    (2, 64 * 96, 1, 60, 4, 5, 96),       # the benchmark's counts: the quad kernel serves them; the flat kernel forced on
])
def test_flat_build_x_equals_the_quad_kernel(gpu_device, monkeypatch, dt, B, N, T_in, F, Fs, Ff, cpad):
    """p4c_build_x for feature counts off the 16-byte grid streams its sources flat through an LDS tile of rows (csrc/rollout.hip:
    build_x_flat_kernel; lightning.py:711-767): bit for bit what the quad kernel writes (P4C_NO_FLAT_STEP=1) -- a conversion per
    element, no arithmetic -- and what torch.cat gives."""
    from py4cast_amd import ops

    g = torch.Generator(device=gpu_device).manual_seed(41)
    prev = torch.randn(B, T_in, N, F, generator=g, device=gpu_device)
    st = torch.rand(B, N, Fs, generator=g, device=gpu_device)
    fo = torch.rand(B, N, Ff, generator=g, device=gpu_device)
    res = {}
    for mode in ("flat", "quad"):
        monkeypatch.setenv("P4C_NO_FLAT_STEP", "1" if mode == "quad" else "0")
        monkeypatch.setenv("P4C_FORCE_FLAT_STEP", "1" if mode == "flat" else "0")
        res[mode] = ops.build_x(prev, st, fo, c_pad=cpad, dtype=dt)
    torch.cuda.synchronize()
    assert torch.equal(_bits(res["flat"]), _bits(res["quad"]))
    want = torch.zeros(B, N, cpad, device=gpu_device)
    want[..., : T_in * F] = prev.permute(0, 2, 1, 3).reshape(B, N, T_in * F)
    want[..., T_in * F: T_in * F + Fs] = st
    want[..., T_in * F + Fs: T_in * F + Fs + Ff] = fo
    assert torch.equal(_bits(res["flat"].reshape(B, N, cpad)), _bits(want.to(dt)))
