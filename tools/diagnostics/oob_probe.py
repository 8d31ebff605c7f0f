"""Out-of-bounds write check of the single-op convolution entry points: every buffer is carved out of one sentinel-filled arena with
guard bands either side; after the launch the guards (and the read-only inputs) must be untouched."""
import sys, torch
sys.path.insert(0, ".")
from py4cast_amd import _lib as L
import py4cast_amd.ops_model as OM
dev = "cuda"
GUARD = 1 << 16   # floats
SENT = 1234.5

class Arena:
    def __init__(self, n):
        self.buf = torch.full((n,), SENT, dtype=torch.float32, device=dev); self.off = GUARD; self.spans = []
    def take(self, numel, dtype=torch.float32):
        nf = (numel * (2 if dtype == torch.bfloat16 else 4) + 3) // 4
        nf = (nf + 63) // 64 * 64
        v = self.buf[self.off:self.off + nf]
        self.spans.append((self.off, self.off + nf)); self.off += nf + GUARD
        return v.view(dtype)[:numel] if dtype != torch.float32 else v[:numel]
    def check(self, tag):
        mask = torch.ones_like(self.buf, dtype=torch.bool)
        for a, b in self.spans:
            mask[a:b] = False
        bad = ((self.buf != SENT) & mask).nonzero().flatten()
        if bad.numel():
            first = int(bad[0]); owner = max((i for i, (a, b) in enumerate(self.spans) if a <= first), default=-1)
            print(f"   !! {tag}: {bad.numel()} guard words overwritten, first at {first} (after span {owner} {self.spans[owner] if owner >= 0 else None}), last {int(bad[-1])}")
        return bad.numel() == 0

def run(B, H, W, CIp, CO, CI, ks, dt):
    comp = "bf16" if dt == torch.bfloat16 else "f32"
    code = L.dtype_code(dt)
    A = Arena(64 << 20)
    x = A.take(B * H * W * CIp, dt).view(B, H, W, CIp); x.copy_(torch.randn(B, H, W, CIp, device=dev).to(dt))
    w = torch.randn(64, CI, ks, ks, device=dev)
    wp_n = 64 * CIp * ks * ks
    wp = A.take(wp_n, dt)
    L.call("p4c_prep_weights", L.ptr(w), 64, CI, ks, 0, 64, CIp, L.ptr(wp), OM._compute(comp), L.stream(x.device))
    ok = A.check("prep_weights")
    out = A.take(B * H * W * 64, dt).view(B, H, W, 64)
    x0 = x.clone()
    L.call("p4c_conv_fwd", L.ptr(x), OM._compute(comp), code, CIp, L.ptr(wp), ks, None, None, 0, None, L.ptr(out), 64, None, B, H, W, 1, L.stream(x.device))
    torch.cuda.synchronize()
    ok &= A.check("conv_fwd"); ok &= bool((x0 == x).all())
    # data gradient: 64 -> CIp (m_blocks)
    mb = (CIp + 63) // 64
    wt = A.take(64 * mb * 64 * ks * ks, dt)
    L.call("p4c_prep_weights", L.ptr(w), 64, CI, ks, 1, 64 * mb, 64, L.ptr(wt), OM._compute(comp), L.stream(x.device))
    dy = A.take(B * H * W * 64, dt).view(B, H, W, 64); dy.copy_(torch.randn(B, H, W, 64, device=dev).to(dt))
    dx = A.take(B * H * W * 64 * mb, dt).view(B, H, W, 64 * mb)
    L.call("p4c_conv_fwd", L.ptr(dy), OM._compute(comp), code, 64, L.ptr(wt), ks, None, None, 0, None, L.ptr(dx), 64 * mb, None, B, H, W, mb, L.stream(x.device))
    torch.cuda.synchronize()
    ok &= A.check("dgrad")
    nbytes = L.lib().p4c_conv_wgrad_workspace_bytes(CIp, ks)
    ws = A.take(nbytes // 4)
    grad = A.take(64 * CI * ks * ks); grad.zero_()
    L.call("p4c_conv_wgrad", L.ptr(x), OM._compute(comp), code, CIp, ks, None, None, 0, L.ptr(dy), 64, CI, L.ptr(grad), L.ptr(ws), B, H, W, L.stream(x.device))
    torch.cuda.synchronize()
    ok &= A.check("wgrad"); ok &= bool((x0 == x).all())
    print(("ok " if ok else "BAD"), B, H, W, CIp, CI, ks, dt)

for dt in (torch.float32, torch.bfloat16):
    for (B, H, W, CIp, CI, ks) in ((2, 64, 96, 32, 16, 3), (2, 64, 96, 32, 13, 1), (2, 16, 24, 32, 32, 3), (2, 8, 12, 64, 64, 3), (2, 64, 96, 64, 64, 3),
                                   (2, 64, 96, 96, 96, 3), (1, 33, 47, 32, 21, 3), (2, 5, 7, 64, 64, 3), (2, 128, 128, 64, 64, 1), (1, 512, 512, 64, 64, 3)):
        run(B, H, W, CIp, 64, CI, ks, dt)
