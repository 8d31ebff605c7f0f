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


def test_in_place_gradients_leave_inside_the_backward_unetrpp():
    """VERDICT r5 item 8 on the real ops: UNETR++'s weight gradients are added into the flat bucket by the TN reductions
    (ops_gemm.GRADS_IN_PLACE) -- FlatDDP(overlap=True) counts the reported writes in its first armed backward and from the second one
    on puts buckets on the communication stream INSIDE the backward, with the gradients of the exchange-after-the-backward run.
    One rank over RCCL with a nominal world size of 2 (the sum over the one rank, halved)."""
    from py4cast_amd.lightning import AutoRegressiveLightning
    from py4cast_amd.trainer import FlatDDP
    from tests.helpers import make_batch, make_dataset_info, synthetic_case

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(31700 + os.getpid() % 1000))
    device = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        H, W, F, Ff, T = 64, 64, 6, 5, 3
        case = synthetic_case(seed=77, B=2, T=T, H=H, W=W, F=F, Ff=Ff, border=0)
        info = make_dataset_info(case, Ff)
        mse = [{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}]
        settings = dict(activation_dtype="bf16", hidden_size=128, num_heads_encoder=2, num_heads_decoder=2, depths=[1, 1, 1, 1],
                        encoder_proj_sizes=[16, 16, 8, 4], decoder_proj_size=16, linear_upsampling=True, attention_code="torch",
                        conv8_dropout=0.0)
        out = {}
        for tag, overlap in (("after", False), ("overlap", True)):
            torch.manual_seed(78)
            lm = AutoRegressiveLightning(settings, info, None, num_input_steps=1, num_pred_steps_train=T, batch_size=2, model_name="UNetRPP",
                                         losses=mse, training_strategy="diff_ar").to(device).train()
            ddp = FlatDDP(lm.model, world_size=2, bucket_bytes=1 << 19, single_bucket_bytes=1 << 16, overlap=overlap)
            assert len(ddp.buckets) > 4 and ddp.overlap == overlap
            runs = []
            for i in range(3):
                ddp.zero_grad()
                loss = lm.training_step(make_batch(case, device), i)
                ddp.arm()
                loss.backward()
                n_in = ddp.issued_in_backward
                ddp.all_reduce_grads()
                torch.cuda.synchronize()
                runs.append((n_in, ddp.flat_grad[: ddp.total].clone()))
            out[tag] = (runs, len(ddp.buckets), sum(ddp.params[i].numel() for i in ddp._inplace), ddp.total)
            ddp.close()
        runs, nb, n_inplace, n_all = out["overlap"]
        assert n_inplace > n_all // 2, (n_inplace, n_all)        # most of the model's gradient bytes bypass autograd
        assert runs[0][0] == 0                                   # the learning pass holds every bucket with such a parameter back
        assert runs[1][0] >= 1 and runs[2][0] >= 1, (runs[1][0], nb)   # then buckets leave before the backward returns
        ref = out["after"][0]
        assert all(r[0] == 0 for r in ref)
        for got, want in zip(runs, ref):                         # same gradients as the exchange after the backward
            assert float(want[1].abs().max()) > 0
            torch.testing.assert_close(got[1], want[1], rtol=1e-5, atol=1e-7)
    finally:
        dist.destroy_process_group()


def _assert_params_close(got, ref, lr=1e-3, steps=3):
    """Adam moves every element by ~lr per step whatever the gradient's size: where the gradient is noise (reduction order differs
    between one process and two ranks) the sign of a step can flip.  So: all but a handful of elements agree closely, and nothing is
    further apart than the steps taken."""
    diff = (got - ref).abs()
    assert float((diff > 2e-5 + 2e-3 * ref.abs()).float().mean()) < 1e-3, float((diff > 2e-5 + 2e-3 * ref.abs()).float().mean())
    assert float(diff.max()) <= 2 * lr * steps + 1e-6, float(diff.max())


def _run(cmd, timeout=600):
    import subprocess

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs of one node (the driver's multi-GPU box)")
@pytest.mark.parametrize("sharded", [False, True])
def test_two_ranks_train_like_one_process_with_the_global_batch(tmp_path, sharded):
    """Two ranks over RCCL (2 samples each) == one process with the 4 samples: identical parameters on both ranks after three
    optimizer steps, equal to the single-process run's, and the same loss sequence (GroupNorm: statistics are per sample, so data
    parallelism changes nothing but the reduction order).  Both exchanges: one all-reduce, and reduce-scatter + sharded AdamW +
    all-gather."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tests", "dist_check.py")
    extra = ["sharded"] if sharded else []
    one = _run([sys.executable, script, str(tmp_path)] + extra)
    assert one.returncode == 0, one.stderr[-2000:]
    port = 32500 + os.getpid() % 1000
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), script, str(tmp_path)] + extra)
    assert two.returncode == 0, two.stderr[-2000:]
    r0, r1 = torch.load(tmp_path / "rank0_w2.pt"), torch.load(tmp_path / "rank1_w2.pt")
    ref = torch.load(tmp_path / "rank0_w1.pt")
    assert torch.equal(r0["params"], r1["params"])
    _assert_params_close(r0["params"], ref["params"])
    torch.testing.assert_close(torch.tensor(r0["losses"]), torch.tensor(ref["losses"]), rtol=1e-4, atol=0)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs of one node (the driver's multi-GPU box)")
def test_bench_runs_on_two_gpus():
    """`python bench.py --gpus 2` (the driver's launch form is the torch.distributed.run line bench.py builds itself): one JSON line
    from rank 0 with n_gpus == 2 and the whole-job aggregate."""
    import json
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = _run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"], 900)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["scaling"] == "weak" and out["value"] > 0


@pytest.mark.parametrize("sharded", [False, True])
def test_two_ranks_sharing_one_gpu_train_like_one_process(tmp_path, sharded):
    """The same check as the RCCL one above on a ONE-GPU box: two ranks, both on cuda:0, exchanging through gloo (device tensors
    staged through the host).  Everything but the transport is the production path -- the native rollout on each rank's half of the
    batch, FlatDDP's buckets on its communication stream, the (sharded) FlatAdamW kernel, the parameter all-gather."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tests", "dist_check.py")
    extra = ["sharded"] if sharded else []
    one = _run([sys.executable, script, str(tmp_path)] + extra)
    assert one.returncode == 0, one.stderr[-2000:]
    port = 33500 + os.getpid() % 1000
    import subprocess

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", P4C_DIST_SHARE_GPU="1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), script, str(tmp_path)] + extra, capture_output=True, text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    r0, r1 = torch.load(tmp_path / "rank0_w2.pt"), torch.load(tmp_path / "rank1_w2.pt")
    ref = torch.load(tmp_path / "rank0_w1.pt")
    assert torch.equal(r0["params"], r1["params"])
    _assert_params_close(r0["params"], ref["params"])
    torch.testing.assert_close(torch.tensor(r0["losses"]), torch.tensor(ref["losses"]), rtol=1e-4, atol=0)


def test_bench_two_ranks_sharing_one_gpu():
    """bench.py's N > 1 path (child job through torch.distributed.run, barrier + synchronize bracketing, MAX over ranks, whole-job
    aggregate, one JSON line from rank 0) on a 1-GPU box: two ranks on cuda:0 over gloo.  Functional only."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", P4C_DIST_SHARE_GPU="1")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--grid", "128", "128",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["scaling"] == "weak" and out["value"] > 0
    assert abs(out["value"] - 2 * 2 * 1e3 / out["ms_per_step"]) < 1e-6 * out["value"]     # whole-job aggregate = ranks x B / step time
    # host safety of the N > 1 path: every rank on its own cores, and ONE launch mode agreed over the ranks
    cores = out["config"]["host_cores_per_rank"]
    if os.cpu_count() >= 2:
        assert len(cores) == 2 and cores[0] and cores[1] and not set(cores[0]) & set(cores[1]), cores
    probe = out["config"]["launch_mode_probe"]
    assert probe is not None and probe["agreed_over_ranks"] == 2 and probe["chosen"] in ("eager", "graph", "graph, single stream")
    assert out["config"]["hip_graph"] == (probe["chosen"] != "eager")


def test_bench_unetrpp_two_ranks_issue_buckets_inside_the_backward():
    """bench.py's N > 1 path for the one model whose gradient exchange is not latency-sized (UNETR++, in-place weight gradients): two
    ranks on cuda:0 over gloo, 8 MB buckets -- after the learning step the buckets are issued from inside the backward.  Functional."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", P4C_DIST_SHARE_GPU="1")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--model", "UNetRPP", "--strategy", "diff_ar", "--hidden", "512",
                          "--steps", "2", "--warmup", "1", "--grid", "128", "128", "--bucket-mb", "8", "--hip-graph", "off", "--no-cpu-baseline", "--no-native-share"],
                         capture_output=True, text=True, timeout=1200, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    gx = out["config"]["gradient_exchange"]
    assert out["n_gpus"] == 2 and gx["buckets"] > 2 and gx["overlapped_with_backward"], gx
    assert gx["buckets_issued_inside_last_backward"] >= 1, gx
