"""Per-launch time of the tall-skinny kernels at the UNetRPP bench shapes (hidden 1024, HEADS heads (default 16: head widths 8 / 16 /
32 / 64; HEADS=4: 32 / 64 / 128 / 256, the widths the VALU kernels took in 64-column chunks) at 16 384 / 4 096 / 1 024 / 256 tokens, B = 2), HIP events over a replayed graph of 20 calls; algorithmic bytes / time against 8 TB/s.
P4C_TS_NO_MFMA=1 times the VALU kernels (in 64-column chunks) for comparison."""
import os, sys, torch
sys.path.insert(0, ".")
from py4cast_amd import ops_ts as TS

dev = torch.device("cuda", 0)
import os
B, H = 2, int(os.environ.get('HEADS', '16'))
rows = []
for N, d, p in [(16384, 128 // H, 64), (4096, 256 // H, 64), (1024, 512 // H, 64), (256, 1024 // H, 32)]:
    qkvv = torch.randn(B, N, 4, H, d, device=dev).to(torch.bfloat16)
    q, k = qkvv[:, :, 0].permute(0, 2, 1, 3), qkvv[:, :, 1].permute(0, 2, 1, 3)
    S = torch.randn(B, N, H, p, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3)
    mdd, mdp, mpd = (torch.randn(B, H, a, b2, device=dev) for a, b2 in ((d, d), (d, p), (p, d)))
    cases = [("apply d->d", lambda: TS._apply_raw(q, mdd, torch.bfloat16), B * H * N * 2 * d * 2),
             ("apply d->p", lambda: TS._apply_raw(q, mdp, torch.bfloat16), B * H * N * (d + p) * 2),
             ("apply p->d", lambda: TS._apply_raw(S, mpd, torch.bfloat16), B * H * N * (d + p) * 2),
             ("gram d,d", lambda: TS._gram_raw(q, k), B * H * N * 2 * d * 2),
             ("gram d,p", lambda: TS._gram_raw(q, S), B * H * N * (d + p) * 2)]
    for name, fn, nbytes in cases:
        # replayed from a HIP graph of 20 calls: device time per call (all its launches), no host in the way
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 100
        print(f"N={N:6d} d={d:3d} p={p:2d} {name:11s} {us:8.1f} us per call (all its launches, graph replay)  {nbytes / us / 1e6:6.2f} TB/s algorithmic  frac {nbytes / us / 1e6 / 8:5.3f}")
