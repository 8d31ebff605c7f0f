"""Is a hipMemsetAsync captured into a HIP graph re-executed by every replay?  (torch's multi-block reductions zero their semaphores with one.)"""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
for nbytes in (4, 64, 256, 4096, 1 << 20):
    n = max(nbytes // 4, 1)
    x = torch.ones(n, device="cuda"); y = torch.zeros(n, device="cuda")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        rc = hip.hipMemsetAsync(ctypes.c_void_p(x.data_ptr()), 0, nbytes, ctypes.c_void_p(st))
        y.copy_(x)
    res = []
    for i in range(3):
        x.fill_(1.0); torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        res.append(float(y.abs().sum()))
    print(f"memset of {nbytes} bytes (rc {rc}): sum(y) after replays 1..3 = {res}   (0 = the memset ran in the replay)")
