"""RCCL smoke test on one GPU: the flat-bucket exchange of FlatDDP through the nccl (= RCCL) backend with a single rank.
The multi-rank arithmetic is covered on CPU (tests/test_ddp_cpu.py, gloo, world_size 2); this checks that the GPU code path --
process-group creation bound to the device, broadcast of the parameters, all-reduce of the flat gradient views, barrier, MAX
reduction of the step time as bench.py does it -- runs on the RCCL build of this image."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_flat_bucket_exchange_over_rccl_single_rank():
    from py4cast_amd.trainer import FlatDDP

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(31500 + os.getpid() % 1000))
    device = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        torch.manual_seed(3)
        net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4)).to(device)
        ddp = FlatDDP(net, world_size=1)
        ddp.broadcast_parameters()
        x = torch.randn(5, 8, device=device)
        net(x).square().mean().backward()
        before = ddp.flat_grad.clone()
        ddp.world_size = 2           # take the exchange path: sum over the (one) rank, divided by the nominal world size
        ddp.all_reduce_grads()
        ddp.world_size = 1
        torch.testing.assert_close(ddp.flat_grad, before / 2)
        assert all(p.grad.data_ptr() >= ddp.flat_grad.data_ptr() for p in net.parameters())
        dist.barrier()
        t = torch.tensor([1.5], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.5
    finally:
        dist.destroy_process_group()
