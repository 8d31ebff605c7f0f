"""world_size-2 gloo test of the flat-bucket gradient exchange used for data-parallel training (CPU)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from py4cast_amd.trainer import FlatDDP

    torch.manual_seed(100 + rank)  # different initial weights on purpose: FlatDDP must broadcast rank 0's
    net = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.Tanh(), torch.nn.Linear(8, 2))
    ddp = FlatDDP(net, world)
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    torch.manual_seed(7 + rank)
    x, y = torch.randn(5, 4), torch.randn(5, 2)
    # two micro-batches accumulate locally, one all-reduce at the optimizer step (accumulate_grad_batches semantics)
    for _ in range(2):
        ((net(x) - y) ** 2).mean().backward()
    local = ddp.flat_grad.clone()
    ddp.all_reduce_grads()
    ret[rank] = (w0, local, ddp.flat_grad.clone(), [p.grad.data_ptr() for p in net.parameters()], ddp.flat_grad.data_ptr())
    dist.destroy_process_group()


def test_flat_ddp_allreduce_mean_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    w_a, local_a, red_a, ptrs_a, base_a = ret[0]
    w_b, local_b, red_b, _, _ = ret[1]
    assert torch.equal(w_a, w_b)  # parameters broadcast from rank 0
    assert not torch.allclose(local_a, local_b)
    torch.testing.assert_close(red_a, (local_a + local_b) / 2)
    torch.testing.assert_close(red_a, red_b)
    assert ptrs_a[0] == base_a  # param.grad tensors are views into the single flat bucket


def _graph_worker(rank, world, port, tmp_dir, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import make_dataset_info, synthetic_case
    from py4cast_amd.lightning import AutoRegressiveLightning

    case = synthetic_case(seed=5, B=2, T=2, H=27, W=27, F=5, Ff=5)
    # every rank constructs the module; only rank 0 builds the mesh graph (rank_zero_setup), the others must find its file
    lm = AutoRegressiveLightning({"tmp_dir": tmp_dir, "processor_layers": 1}, make_dataset_info(case, 5), None, batch_size=2,
                                 model_name="GraphLAM", training_strategy="diff_ar")
    ret[rank] = (lm.model.n_mesh, int(lm.model.g2m_index.shape[1]), sum(p.numel() for p in lm.model.parameters()))
    dist.destroy_process_group()


def test_rank_zero_setup_is_visible_to_all_ranks_gloo(tmp_path):
    """Graph models: rank 0 writes the mesh graph, a barrier makes it visible before the other ranks' constructors read it
    (py4cast/lightning.py:141-144 leaves that ordering to Lightning)."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 30500 + (os.getpid() % 1000)
    mp.spawn(_graph_worker, args=(world, port, str(tmp_path), ret), nprocs=world, join=True)
    assert ret[0] == ret[1] and ret[0][0] == 81


def _bucket_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from py4cast_amd.trainer import FlatDDP

    def make():
        torch.manual_seed(11)
        return torch.nn.Sequential(torch.nn.Linear(37, 101), torch.nn.Tanh(), torch.nn.Linear(101, 53), torch.nn.Tanh(), torch.nn.Linear(53, 3))

    out = {}
    torch.manual_seed(50 + rank)
    x, y = torch.randn(9, 37), torch.randn(9, 3)
    for tag, kw in (("single", {}), ("buckets", dict(bucket_bytes=4096, single_bucket_bytes=1024)),
                    ("sharded", dict(bucket_bytes=4096, single_bucket_bytes=1024, sharded=True))):
        net = make()
        ddp = FlatDDP(net, world, **kw)
        ((net(x) - y) ** 2).mean().backward()
        ddp.all_reduce_grads()
        g = ddp.flat_grad[: ddp.total].clone()
        own = torch.zeros(ddp.flat_grad.numel(), dtype=torch.bool)
        for lo, hi in ddp.shards():
            own[lo:hi] = True
        if tag == "sharded":   # a plain SGD step on the owned shards, then the all-gather makes the parameters whole again
            with torch.no_grad():
                for lo, hi in ddp.shards():
                    ddp.flat_param[lo:hi] -= 0.1 * ddp.flat_grad[lo:hi]
            ddp.all_gather_params()
        out[tag] = (g, own[: ddp.total].clone(), len(ddp.buckets), torch.cat([p.detach().reshape(-1) for p in net.parameters()]))
    ret[rank] = out
    dist.destroy_process_group()


def test_bucketed_and_sharded_exchange_gloo():
    """FlatDDP with several buckets == one bucket; sharded (reduce-scatter, step on the owned shards, all-gather) leaves every rank
    with the parameters a full all-reduce + full step gives."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 1000)
    mp.spawn(_bucket_worker, args=(world, port, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    assert a["single"][2] == 1 and a["buckets"][2] > 3 and a["sharded"][2] > 3
    torch.testing.assert_close(a["single"][0], b["single"][0])
    torch.testing.assert_close(a["buckets"][0], a["single"][0])           # bucketing changes nothing
    for r in (a, b):                                                       # the owned shards hold the mean; the two ranks' shards tile the buffer
        own = r["sharded"][1]
        torch.testing.assert_close(r["sharded"][0][own], a["single"][0][own])
    assert bool((a["sharded"][1] ^ b["sharded"][1]).all())
    # parameters after the sharded SGD step + all-gather == initial - 0.1 * mean gradient, on both ranks
    torch.manual_seed(11)
    init = torch.cat([p.detach().reshape(-1) for p in torch.nn.Sequential(torch.nn.Linear(37, 101), torch.nn.Tanh(), torch.nn.Linear(101, 53),
                                                                         torch.nn.Tanh(), torch.nn.Linear(53, 3)).parameters()])
    torch.testing.assert_close(a["sharded"][3], init - 0.1 * a["single"][0])
    torch.testing.assert_close(b["sharded"][3], a["sharded"][3])


def _overlap_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from py4cast_amd.trainer import FlatDDP

    def make():
        torch.manual_seed(11)
        return torch.nn.Sequential(torch.nn.Linear(24, 96), torch.nn.Tanh(), torch.nn.Linear(96, 64), torch.nn.Tanh(), torch.nn.Linear(64, 24))

    torch.manual_seed(70 + rank)
    x, y = torch.randn(9, 24), torch.randn(9, 3, 24)
    out = {}
    for tag, kw in (("after", dict(overlap=False)), ("overlap", dict(overlap=True)), ("overlap_sharded", dict(overlap=True, sharded=True))):
        net = make()
        ddp = FlatDDP(net, world, bucket_bytes=4096, single_bucket_bytes=1024, **kw)
        events = []
        real = ddp._exchange_bucket
        ddp._exchange_bucket = lambda b, real=real: (events.append(("bucket", b)), real(b))[1]
        for micro in range(2):                       # two accumulated micro-batches: only the second one exchanges
            state, loss = x, 0.0
            for t in range(3):                       # an AR rollout: every parameter is used three times (BPTT)
                state = net(state)
                loss = loss + ((state - y[:, t]) ** 2).mean()
            if micro == 1:
                ddp.arm()
            torch.autograd.backward(loss / 2)
            events.append(("backward_done", micro))
        n_in = ddp.issued_in_backward
        ddp.all_reduce_grads()
        own = torch.zeros(ddp.flat_grad.numel(), dtype=torch.bool)
        for lo, hi in ddp.shards():
            own[lo:hi] = True
        out[tag] = (ddp.flat_grad[: ddp.total].clone(), events, n_in, len(ddp.buckets), own[: ddp.total].clone())
    ret[rank] = out
    dist.destroy_process_group()


def test_gradient_exchange_overlaps_the_backward_gloo():
    """FlatDDP(overlap=True): with BPTT a parameter's gradient is final when autograd has summed its contributions of all AR steps,
    i.e. inside the backward; its bucket is issued right then -- buckets strictly last -> first on every rank -- and not in the
    non-stepping micro-batch; the result equals the exchange issued after the backward (all-reduce and reduce-scatter forms)."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 32500 + (os.getpid() % 1000)
    mp.spawn(_overlap_worker, args=(world, port, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    nb = a["overlap"][3]
    assert nb > 3
    for r in (a, b):
        ev = r["overlap"][1]
        first_done, second_done = ev.index(("backward_done", 0)), ev.index(("backward_done", 1))
        issued = [i for i, e in enumerate(ev) if e[0] == "bucket"]
        assert all(i > first_done for i in issued)                       # nothing in the non-stepping micro-batch
        inside = [ev[i][1] for i in issued if i < second_done]
        assert len(inside) == r["overlap"][2] and len(inside) >= nb - 1  # (the first layer's bucket completes with the sweep itself)
        order = [ev[i][1] for i in issued]
        assert order == sorted(order, reverse=True) and len(order) == nb  # last -> first, each bucket once
        assert [e for e in r["after"][1] if e[0] == "bucket"] and r["after"][2] == 0
        assert r["after"][1].index(("backward_done", 1)) < min(i for i, e in enumerate(r["after"][1]) if e[0] == "bucket")
    torch.testing.assert_close(a["overlap"][0], a["after"][0])
    torch.testing.assert_close(b["overlap"][0], a["after"][0])
    for r in (a, b):
        own = r["overlap_sharded"][4]
        torch.testing.assert_close(r["overlap_sharded"][0][own], a["after"][0][own])
        # sharded AND overlapped: the reduce-scatters too are issued inside the stepping backward, last -> first, each bucket once
        ev = r["overlap_sharded"][1]
        first_done, second_done = ev.index(("backward_done", 0)), ev.index(("backward_done", 1))
        issued = [i for i, e in enumerate(ev) if e[0] == "bucket"]
        assert all(i > first_done for i in issued)
        inside = [ev[i][1] for i in issued if i < second_done]
        assert len(inside) == r["overlap_sharded"][2] and len(inside) >= r["overlap_sharded"][3] - 1
        order = [ev[i][1] for i in issued]
        assert order == sorted(order, reverse=True) and len(order) == r["overlap_sharded"][3]
    assert bool((a["overlap_sharded"][4] ^ b["overlap_sharded"][4]).all())   # the two ranks' shards tile the buffer


def _replaced_grads_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from py4cast_amd.trainer import FlatDDP

    def make():
        torch.manual_seed(11)
        return torch.nn.Sequential(torch.nn.Linear(24, 96), torch.nn.Tanh(), torch.nn.Linear(96, 64), torch.nn.Tanh(), torch.nn.Linear(64, 24))

    torch.manual_seed(90 + rank)
    x, y = torch.randn(9, 24), torch.randn(9, 24)
    out = {}
    for tag in ("views", "set_to_none", "replaced_in_backward"):
        net = make()
        ddp = FlatDDP(net, world, bucket_bytes=4096, single_bucket_bytes=1024, overlap=True)
        opt = torch.optim.SGD(net.parameters(), lr=0.1)
        if tag == "set_to_none":
            opt.zero_grad(set_to_none=True)      # what a host loop that does not know about FlatDDP.zero_grad does
        ddp.arm()
        armed = ddp._armed
        loss = ((net(x) - y) ** 2).mean()
        if tag == "replaced_in_backward":
            # a gradient tensor swapped for another one while the armed backward runs (a hook that assigns p.grad)
            first = next(net.parameters())
            first.register_post_accumulate_grad_hook(lambda p: setattr(p, "grad", p.grad.clone()))
        loss.backward()
        n_in = ddp.issued_in_backward
        ddp.all_reduce_grads()
        grads = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        out[tag] = (grads.clone(), ddp.flat_grad[: ddp.total].clone(), armed, n_in, ddp._grads_are_views())
    # the same replacement with sharded=True.  The hook-issued reduce-scatters have already overwritten this rank's shards
    # with means: a second exchange would mix means with raw sums, so all_reduce_grads must refuse (both ranks, same branch)
    net = make()
    ddp = FlatDDP(net, world, bucket_bytes=4096, single_bucket_bytes=1024, overlap=True, sharded=True)
    ddp.arm()
    loss = ((net(x) - y) ** 2).mean()
    first = next(net.parameters())
    first.register_post_accumulate_grad_hook(lambda p: setattr(p, "grad", p.grad.clone()))
    loss.backward()
    try:
        ddp.all_reduce_grads()
        out["sharded_replaced"] = (ddp.issued_in_backward, None)
    except RuntimeError as exc:
        out["sharded_replaced"] = (ddp.issued_in_backward, str(exc))
    ret[rank] = out
    dist.destroy_process_group()


def test_overlap_with_replaced_gradient_tensors_gloo():
    """ADVICE r3 (trainer.py:157): after ``zero_grad(set_to_none=True)`` (or any ``p.grad = ...``) the gradients are no longer views
    of the flat bucket.  ``arm()`` then stays un-armed -- no hook may reduce the stale flat buffer -- and ``all_reduce_grads`` takes
    the regather path; a tensor replaced DURING an armed backward makes it exchange every bucket again.  Either way every rank ends
    with the same mean the view path gives, and the gradients are views again."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 33500 + (os.getpid() % 1000)
    mp.spawn(_replaced_grads_worker, args=(world, port, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    assert a["views"][2] and a["views"][3] >= 1                    # the plain case is armed and overlaps
    assert not a["set_to_none"][2] and a["set_to_none"][3] == 0    # un-armed: nothing issued from hooks
    for tag in ("set_to_none", "replaced_in_backward"):
        for r in (a, b):
            torch.testing.assert_close(r[tag][0], a["views"][0])   # same mean as the view path, on both ranks
            torch.testing.assert_close(r[tag][1], a["views"][1])
            assert r[tag][4]                                       # p.grad is its slice of the flat bucket again
    torch.testing.assert_close(a["views"][0], b["views"][0])
    for r in (a, b):     # sharded + overlapped + replaced during the backward: refused loudly on every rank, never a silent wrong mean
        n_in, msg = r["sharded_replaced"]
        assert n_in >= 1 and msg is not None and "reduce-scattered" in msg


def _in_place_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from py4cast_amd import _lib as L
    from py4cast_amd.ops_rows import grad_view
    from py4cast_amd.trainer import FlatDDP

    class InPlaceLinear(torch.autograd.Function):
        """The protocol of ops_gemm's GRADS_IN_PLACE nodes on CPU tensors: the .grad views are taken in the forward (grad_view reports
        them), the backward ADDS into them, reports the writes and returns None for the parameters -- no AccumulateGrad, no hook."""

        @staticmethod
        def forward(ctx, x, w, b, gw, gb):
            ctx.gw, ctx.gb = gw, gb
            ctx.save_for_backward(x, w.detach())
            return x @ w.t() + b

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            ctx.gw.add_(dy.t() @ x)
            ctx.gb.add_(dy.sum(0))
            L.grad_written(ctx.gw, ctx.gb)
            return dy @ w, None, None, None, None

    def in_place_linear(m, t, attached):
        """attached: the parameters are inputs of the node (their AccumulateGrad nodes are in the graph and receive undefined
        gradients); not attached: the node sees detached weights, as with a rollout's stand-ins -- nothing but the reported writes
        tells anybody that the gradient arrived."""
        gw, gb = grad_view(m.weight), grad_view(m.bias)
        assert gw is not None and gw is not False and gb is not None and gb is not False
        w, b = (m.weight, m.bias) if attached else (m.weight.detach(), m.bias.detach())
        return InPlaceLinear.apply(t, w, b, gw, gb)

    class Net(torch.nn.Module):
        def __init__(self, in_place):
            super().__init__()
            torch.manual_seed(11)
            self.a, self.b, self.c = torch.nn.Linear(24, 96), torch.nn.Linear(96, 64), torch.nn.Linear(64, 24)
            self.in_place = in_place

        def forward(self, x):
            lin = in_place_linear if self.in_place else (lambda m, t, attached: m(t))
            h = torch.tanh(lin(self.a, x, True))
            h = torch.tanh(self.b(h))            # (the middle layer goes through autograd in both flavours: a mixed model)
            return lin(self.c, h, False)

    torch.manual_seed(70 + rank)
    x, y = torch.randn(9, 24), torch.randn(9, 4, 24)

    def step(net, ddp, T):
        ddp.zero_grad()
        state, loss = x, 0.0
        for t in range(T):                       # BPTT: every parameter is used T times
            state = net(state)
            loss = loss + ((state - y[:, t]) ** 2).mean()
        ddp.arm()
        loss.backward()
        n_in = ddp.issued_in_backward
        ddp.all_reduce_grads()
        return n_in, ddp.flat_grad[: ddp.total].clone()

    out = {}
    ref_net = Net(False)
    ref = FlatDDP(ref_net, world, bucket_bytes=4096, single_bucket_bytes=1024, overlap=False)
    net = Net(True)
    ddp = FlatDDP(net, world, bucket_bytes=4096, single_bucket_bytes=1024, overlap=True)
    out["nb"] = len(ddp.buckets)
    out["ref3"] = step(ref_net, ref, 3)[1]
    out["ref2"] = step(ref_net, ref, 2)[1]
    out["learn"] = step(net, ddp, 3)             # first armed backward: counts the in-place writes, holds their buckets back
    out["steady"] = step(net, ddp, 3)            # from now on the buckets leave inside the backward
    out["profile"] = [ddp._profile[i] for i in range(len(ddp.params))if i in ddp._profile]
    out["shorter"] = step(net, ddp, 2)           # fewer uses than learned: nothing leaves early, the result is still the mean
    out["shorter_again"] = step(net, ddp, 2)     # ... and the new counts have been learned
    try:                                         # more uses than learned: a bucket would have left before its last contribution
        step(net, ddp, 3)
        out["longer"] = None
    except RuntimeError as exc:
        out["longer"] = str(exc)
    ret[rank] = out
    dist.destroy_process_group()


def test_in_place_gradients_join_the_overlapped_exchange_gloo():
    """gradients that a kernel adds straight into ``.grad`` (GRADS_IN_PLACE) fire no per-parameter hook.  The ops
    report their writes (``_lib.grad_written``), FlatDDP learns the number of writes per parameter in the first armed backward and
    from the second one on issues the buckets INSIDE the backward (>= 1 before it returns), with the same mean as the exchange after
    the backward; a step with fewer uses falls back to the exchange after the backward, one with more uses than learned raises."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 34500 + (os.getpid() % 1000)
    mp.spawn(_in_place_worker, args=(world, port, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    for r in (a, b):
        assert r["nb"] > 3
        assert r["learn"][0] == 0                                  # the last layer's gradients are in place: nothing may leave yet
        assert r["steady"][0] >= r["nb"] - 1                       # all but (at most) the first layer's bucket inside the backward
        # three writes per in-place parameter (whether the engine also runs the AccumulateGrad node of a parameter whose gradients
        # all came back undefined -- and with it the hook -- is the engine's business: the count is learned, not assumed)
        assert len(r["profile"]) == 4 and all(w == 3 and h in (0, 1) for w, h in r["profile"]) and r["profile"][2:] == [(3, 0), (3, 0)]
        assert r["shorter"][0] == 0 and r["shorter_again"][0] >= r["nb"] - 1
        assert r["longer"] is not None and "after its bucket had been issued" in r["longer"]
        torch.testing.assert_close(r["learn"][1], a["ref3"])
        torch.testing.assert_close(r["steady"][1], a["ref3"])
        torch.testing.assert_close(r["shorter"][1], a["ref2"])
        torch.testing.assert_close(r["shorter_again"][1], a["ref2"])
    torch.testing.assert_close(a["ref3"], b["ref3"])


def test_rank_pinning_hands_out_whole_physical_cores_per_numa_node(monkeypatch):
    """pin_rank_to_cores on an SMT host whose CPU ids run socket 0, socket 1, socket-0 siblings, socket-1 siblings: every
    rank gets whole physical cores (both hardware threads), ranks 0..N/2-1 on node 0 and the rest on node 1, no CPU twice; a rank the
    launcher has already bound (inherited affinity narrower than the machine) is left alone; P4C_NO_AFFINITY=1 likewise."""
    import os

    from py4cast_amd import trainer as T

    ncore = 16                                        # 2 sockets x 8 cores x 2 threads = 32 CPUs
    topo = {}
    for cpu in range(32):
        core = cpu % ncore
        topo[cpu] = (core // 8, core // 8, core % 8)  # (numa node, package, core id): cpu and cpu + 16 are siblings
    pinned = {}
    monkeypatch.setattr(T, "_cpu_topology", lambda cpus: {c: topo[c] for c in cpus})
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(32)))
    monkeypatch.setattr(os, "cpu_count", lambda: 32)
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: pinned.__setitem__("cpus", sorted(cpus)))
    monkeypatch.setattr(T.torch, "set_num_threads", lambda n: None)
    monkeypatch.delenv("P4C_NO_AFFINITY", raising=False)
    seen = []
    for r in range(4):
        mine = T.pin_rank_to_cores(r, 4)
        assert mine == pinned["cpus"] and len(mine) == 8
        assert all((c + 16) % 32 in mine for c in mine)                      # both hardware threads of every core
        assert len({topo[c][0] for c in mine}) == 1 and topo[mine[0]][0] == r // 2
        assert "physical cores per rank" in T.AFFINITY_POLICY[0]
        seen += mine
    assert sorted(seen) == list(range(32))
    # contiguous id slices would have given rank 0 the cpus 0..7 WITHOUT their siblings 16..23
    assert T.pin_rank_to_cores(0, 4) == [0, 1, 2, 3, 16, 17, 18, 19]
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(8)))   # bound by the launcher
    assert T.pin_rank_to_cores(1, 4) is None and "launcher bound" in T.AFFINITY_POLICY[0]
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(32)))
    monkeypatch.setenv("P4C_NO_AFFINITY", "1")
    assert T.pin_rank_to_cores(1, 4) is None and "P4C_NO_AFFINITY" in T.AFFINITY_POLICY[0]
    assert T.pin_rank_to_cores(0, 1) is None
