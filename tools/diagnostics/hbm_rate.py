"""HBM rates this box sustains with library kernels (context for the roofline fractions in DESIGN.md):
device-to-device copy (read + write), fill (write only) and a sum reduction (read only) over 1 GiB, HIP-event timed."""
import torch

assert torch.cuda.is_available()
dev = torch.device("cuda", 0)
n = 1 << 28  # fp32 elements = 1 GiB
x = torch.randn(n, device=dev)
y = torch.empty_like(x)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


gib = n * 4
for name, fn, nbytes in (("copy (read + write)", lambda: y.copy_(x), 2 * gib), ("fill (write)", lambda: y.fill_(1.0), gib),
                         ("sum (read)", lambda: x.sum(), gib), ("bf16 add (2 reads + write)", None, 0)):
    if fn is None:
        xb, yb = x[: n // 2].view(torch.bfloat16), y[: n // 2].view(torch.bfloat16)
        zb = torch.empty_like(xb)
        fn, nbytes = (lambda: torch.add(xb, yb, out=zb)), 3 * xb.numel() * 2
    t = timed(fn)
    print(f"{name:32s} {t * 1e6:9.1f} us  {nbytes / t / 1e12:6.2f} TB/s")
