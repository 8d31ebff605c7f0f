"""Does an EXTERNAL event recorded inside a captured HIP graph release a stream outside the graph while the replay is still running?
(What an exchange overlapped with a REPLAYED backward needs: FlatDDP would record one such event per bucket during the capture and make
its communication stream wait for them after each graph launch.)  Prints what happened; no product code depends on it."""
import sys
import time

import torch

dev = torch.device("cuda:0")
try:
    ev = torch.cuda.Event(external=True)
except TypeError as exc:
    print("external events: not in this torch build:", exc)
    sys.exit(0)
main, side = torch.cuda.Stream(), torch.cuda.Stream()
a = torch.zeros(1 << 24, device=dev)
b = torch.zeros(1 << 26, device=dev)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=main):
        a.add_(1.0)
        ev.record()
        for _ in range(100):      # a long tail after the record (~100 x 0.1 ms)
            b.add_(1.0)
except Exception as exc:        # noqa: BLE001
    print("capture with an external event record failed:", type(exc).__name__, str(exc)[:300])
    sys.exit(0)
for rep in range(3):
    t0, t_side, t_main = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(main):
        t0.record()
        g.replay()
        t_main.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        c = a.clone()
        t_side.record()
    torch.cuda.synchronize()
    print(f"replay {rep}: a seen by the side stream = {float(c[0])} (expected {rep + 1}.0), side done at {t0.elapsed_time(t_side):.3f} ms, "
          f"graph done at {t0.elapsed_time(t_main):.3f} ms")
