"""Does a large column reduction (the bias gradient of a Linear: dy.sum(0) over 10^5-10^6 rows) replay correctly from a HIP graph
on this stack?  (Round 2: SwinUNetR / UNetRPP bias gradients came back as garbage from replays at 512x512 only.)"""
import torch
dev = torch.device("cuda:0")
for rows, cols, dt in [(65536, 96, torch.bfloat16), (262144, 96, torch.bfloat16), (262144, 24, torch.bfloat16), (262144, 96, torch.float32), (524288, 72, torch.bfloat16)]:
    x = torch.randn(rows, cols, device=dev).to(dt)
    ref = x.float().sum(0)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            y = x.sum(0)
            z = (torch.ones(1, rows, device=dev, dtype=dt) @ x)[0]
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = x.sum(0)
        z = (torch.ones(1, rows, device=dev, dtype=dt) @ x)[0]
    outs = []
    for _ in range(3):
        g.replay(); torch.cuda.synchronize()
        outs.append((float((y.float() - ref).abs().max()), float((z.float() - ref).abs().max())))
    print(rows, cols, dt, "eager-vs-replay max abs err (sum, ones@x):", outs)
