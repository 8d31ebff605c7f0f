#!/usr/bin/env python3
"""
bench.py -- AR-training samples/sec on the synthetic 512x512x60 grid, 3-step rollout
(BASELINE.json metric; SURVEY.md section 8(d) workload).

    python bench.py --gpus N --steps K --warmup W [--model HalfUNet] [--dtype bf16|f32]

One "step" = one micro-batch of B samples per GPU through: 3 AR steps (build x -> model forward
-> scaled residual update + border forcing -> weighted MSE) -> backward through the rollout
(BPTT) -> gradient all-reduce (N>1, RCCL) -> AdamW step.  Inputs are synthetic, generated on
the device and resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

For N>1 launch with torch.distributed.run (one rank per GPU); samples are independent, each rank
draws its own B samples (weak scaling) and the only collective is the gradient all-reduce.
"""

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}  # dense peaks, MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--accumulate", type=int, default=1,
                    help="accumulate_grad_batches (trainer.yaml:58): micro-batches per gradient all-reduce + optimizer step; "
                         "a bench step is one micro-batch")
    ap.add_argument("--kernel-times", action="store_true",
                    help="also record HIP events around every C entry point and every tagged conv / weight-gradient launch "
                         "(kernel_ms, data-gradient and weight-gradient launch times); costs ~0.5 ms per step of event markers")
    ap.add_argument("--setup-steps", type=int, default=2,
                    help="untimed one-time initialisation before the W warmup steps (code-object load, workspace allocation, "
                         "RCCL communicator creation); reported as config.setup_steps")
    ap.add_argument("--model", default=os.environ.get("P4C_BENCH_MODEL", "HalfUNet"))
    ap.add_argument("--dtype", default=os.environ.get("P4C_BENCH_DTYPE", "bf16"), choices=["f32", "bf16"],
                    help="matrix-core input type of the model convolutions")
    ap.add_argument("--act-dtype", default=os.environ.get("P4C_BENCH_ACT_DTYPE"), choices=["f32", "bf16"],
                    help="HBM storage of activations (default: same as --dtype)")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--grid", type=int, nargs=2, default=[512, 512])
    ap.add_argument("--features", type=int, default=60)
    ap.add_argument("--forcings", type=int, default=5,
                    help="forcing features per step (5 = the synthetic workload of SURVEY 8d; the shipped Titan configuration has 21: "
                         "--grid 512 640 --features 21 --forcings 21, config/CLI/dataset/titan.yaml:32,38-76)")
    ap.add_argument("--pred-steps", type=int, default=3)
    ap.add_argument("--border", type=int, default=0)
    ap.add_argument("--strategy", default="scaled_ar", choices=["scaled_ar", "diff_ar"],
                    help="training_strategy (BASELINE configuration 5 -- UNetRPP -- is quoted on 6-step diff_ar: --model UNetRPP "
                         "--strategy diff_ar --pred-steps 6)")
    ap.add_argument("--no-native-share", action="store_true", help="skip the extra profiled step that measures the share of GPU kernel "
                    "time spent in this repository's kernels")
    ap.add_argument("--hidden", type=int, default=1024, help="UNetRPP hidden_size (config/CLI/model/unetrpp.yaml:20)")
    ap.add_argument("--unetrpp-block", default="published", choices=["published", "published-nodrop", "restated"],
                    help="UNetRPP transformer block: as published / as mfai wraps it (default: x_SA merged by permute(0,3,1,2).reshape, conv8 = "
                         "Sequential(Dropout2d(0.1), Conv), E = F), the same with the dropout off, or the restated block of rounds 2-5")
    ap.add_argument("--hip-graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the micro-batch (rollout + loss + backward) from a HIP graph; auto: only for models that ask for it "
                         "(launch-bound small-kernel models); the roofline object is then measured in eager steps before the timed region")
    ap.add_argument("--sharded", default="auto", choices=["auto", "on", "off"],
                    help="N > 1: reduce-scatter + sharded AdamW + all-gather instead of all-reduce + full AdamW (auto: above 16 MB of gradients)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: exchange the gradients after the backward instead of inside it")
    ap.add_argument("--bucket-mb", type=int, default=64, help="N > 1: size of the gradient buckets above the 16 MB single-bucket threshold")
    ap.add_argument("--deterministic", action="store_true",
                    help="library convolutions on their deterministic solvers (torch.backends.cudnn.deterministic): SwinUNetR / UNetRPP "
                         "steps then reproduce bit for bit (DESIGN.md 7a)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="CPU-baseline budget: 3 untimed iterations, then iterations are timed until it is spent, at least five")
    ap.add_argument("--cpu-crop", type=int, default=0, help="debugging: time the CPU baseline on a crop of this size (scaled)")
    ap.add_argument("--no-fp32-flavour", action="store_true",
                    help="skip the short fp32 (parity flavour) measurement reported as `fp32_flavour`")
    ap.add_argument("--no-larger-batch", action="store_true", help="skip the short `larger_batch` measurement (bf16, --larger-batch samples per GPU)")
    ap.add_argument("--larger-batch", type=int, default=8)
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default HalfUNet run on one GPU: do not also time BASELINE configurations 3 / 4 / 5 (SwinUNetR, HiLAM, UNetRPP 6-step "
                         "diff_ar; 5 steps each in child processes of their own) into `other_configs`")
    ap.add_argument("--other-configs-launcher", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


# BASELINE.json configurations 3 / 4 / 5 at the bench size, as `bench.py --model ...` runs them (name -> extra flags)
OTHER_CONFIGS = {
    "SwinUNetR": [],                                                  # configuration 3: 3-step scaled_ar
    "HiLAM": [],                                                      # configuration 4: mesh GNN, 3-step scaled_ar
    "UNetRPP": ["--strategy", "diff_ar", "--pred-steps", "6"],        # configuration 5: 6-step diff_ar, hidden 1024
}


def other_configs_launcher():
    """Runs in a child process that the default run starts BEFORE its own first GPU call and that never touches the GPU itself: waits
    for a line on stdin (the headline legs are done), then runs one `bench.py --model X` per configuration as a child of its own -- a
    fresh process each, nothing re-executed from a process that has initialised the GPU -- and prints ONE JSON object with their
    ms_per_step / native_share / roofline.step.frac."""
    import subprocess

    if not sys.stdin.readline().strip():
        return
    args = parse_args()
    size = ["--grid", str(args.grid[0]), str(args.grid[1]), "--batch", str(args.batch), "--features", str(args.features), "--forcings",
            str(args.forcings), "--hidden", str(args.hidden)]      # (the bench size unless the caller shrank it: tests)
    res = {}
    for name, extra in OTHER_CONFIGS.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--model", name, "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
               "--no-other-configs"] + size + extra
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                res[name] = {"error": f"exit code {p.returncode}: {p.stderr.strip()[-400:]}"}
                continue
            d = json.loads(lines[-1])
            roof = d.get("roofline") or {}
            res[name] = {"ms_per_step": d["ms_per_step"], "samples_per_s": d["value"], "steps": d["steps"],
                         "native_share": (d.get("native_share") or {}).get("of_gpu_kernel_time"),
                         "roofline_step_frac": (roof.get("step") or {}).get("frac"), "roofline_kernel": roof.get("kernel"),
                         "roofline_frac": roof.get("frac"), "hip_graph": d["config"].get("hip_graph"), "workload": d["config"]["workload"]}
        except Exception as exc:  # noqa: BLE001  (a configuration that fails is reported as such, the headline line stays valid)
            res[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    print(json.dumps(res), flush=True)


def synthetic_case(seed, B, T, T_in, H, W, F, Ff, Fs, border, device):
    """N(0,1) states clipped to [-3,3], U(0,1) date forcings, U(0,1366) TOA radiation."""
    g = torch.Generator(device=device).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, device=device)
    ru = lambda *s: torch.rand(*s, generator=g, device=device)
    forcing = ru(B, T, H, W, Ff)
    forcing[..., -1] *= 1366.0
    ys, xs = torch.meshgrid(torch.linspace(0, 1, H, device=device), torch.linspace(0, 1, W, device=device), indexing="ij")
    statics = torch.stack([xs, ys, ru(H, W), torch.zeros(H, W, device=device)], dim=-1)[..., :Fs]
    bm = torch.zeros(H, W, 1, device=device)
    if border > 0:
        bm[:border], bm[-border:], bm[:, :border], bm[:, -border:] = 1, 1, 1, 1
    statics[..., 3:4] = bm
    levels = [250, 500, 700, 850]
    sw = torch.tensor([1 + levels[i % 4] / 1000 if i < F - 4 else 2.0 for i in range(F)])
    return dict(
        inputs=rn(B, T_in, H, W, F).clamp(-3, 3),
        forcing=forcing,
        outputs=rn(B, T, H, W, F).clamp(-3, 3),
        statics=statics,
        border_mask=bm,
        diff_std=(ru(F) + 0.5).cpu(),
        diff_mean=(rn(F) * 0.01).cpu(),
        std=(ru(F) + 0.5).cpu(),
        state_weight=sw,
    )


def make_info(case, Ff):
    from py4cast_amd.base import DatasetInfo, Statics, Stats
    from py4cast_amd.namedtensor import NamedTensor

    F = case["diff_std"].shape[0]
    names = [f"f{i}" for i in range(F)]
    gs = NamedTensor(case["statics"].cpu(), ["lat", "lon", "features"], ["x", "y", "geopotential", "border_mask"])
    stats = Stats({n: {"std": case["std"][i], "mean": torch.tensor(0.0)} for i, n in enumerate(names)})
    dstats = Stats({n: {"std": case["diff_std"][i], "mean": case["diff_mean"][i]} for i, n in enumerate(names)})
    return DatasetInfo("synthetic", Statics(gs, tuple(gs.tensor.shape[:2])), stats, dstats,
                       {n: float(case["state_weight"][i]) for i, n in enumerate(names)}, {"input_output": names}, F, Ff)


def model_settings(model, dtype, act_dtype=None, hidden=1024, unetrpp_block="published"):
    """The settings dict of each registry model as the reference's yaml files give it (config/CLI/model/*.yaml), with the
    compute / activation types of this run; shared with tests/test_bench_size_gpu.py."""
    name = model.lower()
    if name.startswith(("graphlam", "hilam")):
        return {"activation_dtype": act_dtype or dtype, "tmp_dir": os.environ.get("TMPDIR", "/tmp")}
    if name.startswith("unetrpp"):   # config/CLI/model/unetrpp.yaml:19-35
        return {"hidden_size": hidden, "num_heads_encoder": 16, "num_heads_decoder": 4, "depths": [3, 3, 3, 3],
                "linear_upsampling": True, "downsampling_rate": 4, "decoder_proj_size": 64, "encoder_proj_sizes": [64, 64, 64, 32],
                "attention_code": "torch", "activation_dtype": act_dtype or dtype,
                # the transformer block as published / as mfai wraps it (x_SA merge, conv8 = Sequential(Dropout2d(0.1), Conv), E = F), or
                # the restated block of rounds 2-5 (py4cast_amd/unetrpp.py); "published-nodrop": the published block with p = 0
                "published_block": unetrpp_block != "restated", "conv8_dropout": 0.1 if unetrpp_block == "published" else 0.0}
    if name.startswith("swin"):
        return {"activation_dtype": act_dtype or dtype}
    if model not in ("Identity",):
        return {"compute_dtype": dtype, "activation_dtype": act_dtype or dtype}
    return {}


def make_batch(case):
    from py4cast_amd.base import ItemBatch
    from py4cast_amd.namedtensor import NamedTensor

    dims = ["batch", "timestep", "lat", "lon", "features"]
    F, Ff = case["inputs"].shape[-1], case["forcing"].shape[-1]
    return ItemBatch(
        NamedTensor(case["inputs"], dims, [f"f{i}" for i in range(F)]),
        NamedTensor(case["forcing"], dims, [f"g{i}" for i in range(Ff)]),
        NamedTensor(case["outputs"], dims, [f"f{i}" for i in range(F)]),
    )


def cpu_baseline(args, seconds):
    """
    The CPU restatement (oracle/, kind "port") of the same step -- rollout + weighted MSE + backward +
    AdamW with the oracle's torch-native model -- timed on the host cores on a bounded sample: ONE sample of the
    FULL grid (no crop, no scaling), three untimed iterations then at least five timed ones (about 4 s each for HalfUNet on 64 cores).
    """
    from oracle import losses as olosses
    from oracle import rollout as orollout

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 64)  # torch intra-op threading stops scaling (and oversubscribes small boxes) beyond this
    torch.set_num_threads(cores)
    H, W = args.grid
    if args.cpu_crop:   # debugging aid only: the default times the full grid
        H, W = min(H, args.cpu_crop), min(W, args.cpu_crop)
    F, T = args.features, args.pred_steps
    Ff = args.forcings
    case = synthetic_case(99, 1, T, 1, H, W, F, Ff, 4, args.border, torch.device("cpu"))
    interior = 1.0 - case["border_mask"]
    wts = olosses.weighted_loss_weights(case["state_weight"], case["diff_std"], "mse")
    statics = case["statics"].unsqueeze(0)
    if args.model == "Identity":
        scaler = torch.rand(1, requires_grad=True)
        params = [scaler]
        model_fn = lambda x: x[..., :F] * scaler
        features_second = False
    elif args.model.lower().startswith("graphlam"):
        from oracle.graphlam import GraphLam as OracleGraphLam
        from py4cast_amd.graph_build import build_mesh_graph   # host-side graph construction (data for the oracle)

        ys, xs = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
        mg = build_mesh_graph(torch.stack([xs, ys]))
        graph = {"g2m": mg.g2m, "m2m": mg.m2m, "m2g": mg.m2g, "g2m_feat": mg.g2m_feat, "m2m_feat": mg.m2m_feat,
                 "m2g_feat": mg.m2g_feat, "mesh_pos": mg.mesh_pos}
        net = OracleGraphLam(F + 4 + Ff, F, graph)
        params = list(net.parameters())
        flat = lambda t: t.flatten(2, 3) if t.dim() == 5 else t.flatten(-3, -2)  # noqa: E731
        case = {k: (flat(v) if k in ("inputs", "forcing", "outputs") else v) for k, v in case.items()}
        case["statics"], case["border_mask"] = case["statics"].flatten(0, 1), case["border_mask"].flatten(0, 1)
        interior, statics = 1.0 - case["border_mask"], case["statics"].unsqueeze(0)
        model_fn = net
        features_second = False
    elif args.model.lower().startswith("hilam"):
        from oracle.hilam import HiLam as OracleHiLam, HiLamParallel as OracleHiLamParallel
        from py4cast_amd.graph_build import build_hierarchical_graph   # host-side graph construction (data for the oracle)

        ys, xs = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
        hg = build_hierarchical_graph(torch.stack([xs, ys]))
        graph = {"g2m": hg.g2m, "m2g": hg.m2g, "g2m_feat": hg.g2m_feat, "m2g_feat": hg.m2g_feat, "mesh_pos": hg.mesh_pos,
                 "same": hg.same, "same_feat": hg.same_feat, "up": hg.up, "up_feat": hg.up_feat, "down": hg.down,
                 "down_feat": hg.down_feat}
        net = (OracleHiLamParallel if "Parallel" in args.model else OracleHiLam)(F + 4 + Ff, F, graph)
        params = list(net.parameters())
        flat = lambda t: t.flatten(2, 3) if t.dim() == 5 else t.flatten(-3, -2)  # noqa: E731
        case = {k: (flat(v) if k in ("inputs", "forcing", "outputs") else v) for k, v in case.items()}
        case["statics"], case["border_mask"] = case["statics"].flatten(0, 1), case["border_mask"].flatten(0, 1)
        interior, statics = 1.0 - case["border_mask"], case["statics"].unsqueeze(0)
        model_fn = net
        features_second = False
    elif args.model.lower().startswith("unetrpp"):
        from oracle.unetrpp import UNetRPP as OracleUNetRPP

        net = OracleUNetRPP(F + 4 + Ff, F, (H, W), hidden_size=args.hidden, published_block=args.unetrpp_block != "restated",
                            conv8_dropout=0.1 if args.unetrpp_block == "published" else 0.0)   # unetrpp.yaml:19-35 defaults otherwise
        params = list(net.parameters())
        model_fn = net
        features_second = False
    elif args.model.lower().startswith("swin"):
        from oracle.swinunetr import SwinUNetR as OracleSwin

        net = OracleSwin(F + 4 + Ff, F)
        params = list(net.parameters())
        model_fn = net
        features_second = False
    else:
        from oracle import halfunet as ohalf

        net = ohalf.HalfUNetRef(F + 4 + Ff, F)
        params = list(net.parameters())
        model_fn = net
        features_second = True
    opt = torch.optim.AdamW(params, lr=1e-3, betas=(0.9, 0.95))

    def one():
        pred = orollout.rollout(model_fn, case["inputs"], case["forcing"], case["outputs"], statics, case["border_mask"],
                                interior, case["diff_std"], case["diff_mean"], args.strategy, 1, False, "train",
                                features_second=features_second)
        loss = olosses.training_loss(pred, case["outputs"], False, [("WeightedLoss", 1.0, dict(weights=wts, interior_mask=interior, kind="mse"))])
        loss.backward()
        opt.step()
        opt.zero_grad()

    # BASELINE.md section 3 / SURVEY 8d: >= 3 warm-up + >= 5 timed iterations.  A bound keeps the default run within minutes on a
    # slow host: when the first iteration alone takes more than a quarter of the budget, the counts shrink (and the JSON says so).
    t1 = time.perf_counter()
    one()
    first = time.perf_counter() - t1
    n_warm, n_timed = (3, 5) if first * 8 <= max(seconds, 1.0) * 4 else (1, 2)
    for _ in range(n_warm - 1):
        one()
    t0, n, times = time.perf_counter(), 0, []
    while n < n_timed or (time.perf_counter() - t0 < seconds and n < 50):
        t1 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t1)
        n += 1
    dt = (time.perf_counter() - t0) / n
    scale = (H * W) / float(args.grid[0] * args.grid[1])
    what = f"the full {H}x{W}x{F} grid" if scale == 1.0 else f"a {H}x{W}x{F} crop, scaled by pixel count to {args.grid[0]}x{args.grid[1]}"
    return {
        "value": scale / dt,
        "unit": "samples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"oracle (torch CPU, fp32) {args.model} training step (T={T} rollout + loss + backward + AdamW) on 1 sample of "
                  f"{what}: {n_warm} untimed + {n} timed iterations, {min(times):.2f}-{max(times):.2f} s each"
                  + ("" if n_warm >= 3 else f" (fewer than 3 + 5: the first iteration took {first:.1f} s of a {seconds:.0f} s budget)"),
    }


def fp32_flavour(args, case, info, device, steps=6, warmup=2, dtype="f32", batch=None):
    from py4cast_amd import _lib as L
    from py4cast_amd.lightning import AutoRegressiveLightning
    from py4cast_amd.trainer import FlatDDP

    H, W = args.grid
    B, T = batch or args.batch, args.pred_steps
    torch.manual_seed(1234)
    lm = AutoRegressiveLightning(
        {"compute_dtype": dtype, "activation_dtype": dtype}, info, None, num_input_steps=1, num_pred_steps_train=T,
        num_pred_steps_val_test=T, batch_size=B, model_name=args.model,
        losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
        training_strategy="scaled_ar", learning_rate=1e-3, min_learning_rate=3e-7, num_warmup_steps=1000, betas=(0.9, 0.95),
    ).to(device)
    ddp = FlatDDP(lm.model, 1)
    opt = lm.configure_optimizers()["optimizer"]

    def step(i):
        loss = lm.training_step(make_batch(case), i)
        loss.backward()
        opt.step()
        ddp.zero_grad()
        return loss

    for i in range(2 + warmup):
        step(i)
    L.lib().p4c_prof_enable(1, min(steps, 4) * T * 3)
    L.lib().p4c_prof_filter(B * H * W)
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    import gc

    gc.collect()
    gc_was_enabled = gc.isenabled()
    gc.disable()   # (see main(): no collector pauses inside a ~0.1 s timed region)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        loss = step(warmup + i).detach()
        marks[i + 1].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if gc_was_enabled:
        gc.enable()
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    roof = lm.model.roofline({}, B=B, H=H, W=W)
    L.lib().p4c_prof_enable(0, 0)
    return {"value": B * steps / dt, "unit": "samples/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
            "step_ms": {"min": step_ms[0], "median": step_ms[len(step_ms) // 2], "max": step_ms[-1]},
            "dtype": dtype, "batch_per_gpu": B, "loss": float(loss.detach()), "roofline": roof}


def native_share_of_one_step(step_fn):
    """Share of GPU kernel time spent in this repository's kernels (names under p4c::) in ONE extra eager step, from the profiler's
    device-side kernel records (roctracer through torch.profiler); the rest is library code (ATen element-wise / reductions / copies,
    hipBLASLt, MIOpen).  Returns (share, {category: ms}, kernel count) or None when the profiler records no device activity."""
    from torch.profiler import ProfilerActivity, profile

    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step_fn()
        torch.cuda.synchronize()
    cat, n = {}, 0
    for ev in prof.events():
        t = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
        if not t or getattr(ev, "device_type", None) is None or "DeviceType.CUDA" not in str(ev.device_type):
            continue
        name = ev.name
        k = ("native" if ("p4c" in name) else "hipBLASLt" if name.startswith("Cijk") else
             "MIOpen" if any(w in name for w in ("MIOpen", "Im2d2Col", "Col2Im", "SubTensorOp", "miopen", "igemm", "naive_conv")) else
             "ATen" if "at::" in name else "other")
        cat[k] = cat.get(k, 0.0) + t / 1e3
        n += 1
    tot = sum(cat.values())
    if n == 0 or tot <= 0:
        return None
    return cat.get("native", 0.0) / tot, {k: round(v, 3) for k, v in sorted(cat.items())}, n


def main():
    args = parse_args()
    if args.other_configs_launcher:
        return other_configs_launcher()
    others = None
    # (never under a profiler: its preloaded library initialises the GPU before this program starts, and a child started from such a
    # process is exactly the re-exec the GPU boxes forbid)
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and args.model == "HalfUNet" and not args.no_other_configs and not args.kernel_times
            and args.dtype == "bf16" and not profiled):
        # the default run also reports configurations 3 / 4 / 5: their launcher is started HERE, before this process touches the GPU
        import subprocess

        others = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--other-configs-launcher", "--grid", str(args.grid[0]),
                                   str(args.grid[1]), "--batch", str(args.batch), "--features", str(args.features), "--forcings",
                                   str(args.forcings), "--hidden", str(args.hidden)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    try:
        _main(args, others)
    finally:
        if others is not None and others.poll() is None:
            try:
                others.stdin.close()      # an empty line: the launcher leaves without running anything
            except OSError:
                pass
            others.wait(timeout=30)


def _main(args, others=None):
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # called as plain `python bench.py --gpus N`: start the N ranks as a child job (nothing has touched the GPU yet in this
        # process) and leave with its exit code
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # each rank of the node on its own block of host cores, before anything of this process touches the GPU (trainer.pin_rank_to_cores)
    from py4cast_amd.trainer import pin_rank_to_cores

    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    affinity = pin_rank_to_cores(local_rank, local_world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    # P4C_DIST_SHARE_GPU=1 (tests on a 1-GPU box only): every rank on cuda:0, gloo transport -- exercises the N > 1 code path of this
    # script (barriers, MAX over ranks, aggregate value); the number it prints is NOT a scaling measurement
    share_gpu = os.environ.get("P4C_DIST_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.deterministic:
        torch.backends.cudnn.deterministic = True
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from py4cast_amd import _lib as L
    from py4cast_amd.lightning import AutoRegressiveLightning
    from py4cast_amd.trainer import FlatDDP

    H, W = args.grid
    B, F, T, Ff, Fs = args.batch, args.features, args.pred_steps, args.forcings, 4
    case = synthetic_case(1234 + rank, B, T, 1, H, W, F, Ff, Fs, args.border, device)
    info = make_info(case, Ff)
    settings = model_settings(args.model, args.dtype, args.act_dtype, args.hidden, args.unetrpp_block)
    torch.manual_seed(1234)  # identical initial weights on every rank
    lm = AutoRegressiveLightning(
        settings, info, None, num_input_steps=1, num_pred_steps_train=T, num_pred_steps_val_test=T, batch_size=B,
        model_name=args.model,
        losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
        training_strategy=args.strategy, learning_rate=1e-3, min_learning_rate=3e-7, num_warmup_steps=1000,
        betas=(0.9, 0.95),
    ).to(device)
    n_grad_bytes = 4 * sum(p.numel() for p in lm.model.parameters() if p.requires_grad)
    sharded = world > 1 and (args.sharded == "on" or (args.sharded == "auto" and n_grad_bytes > (16 << 20)))
    ddp = FlatDDP(lm.model, world, bucket_bytes=args.bucket_mb << 20, sharded=sharded, overlap=not args.no_overlap)
    opt = lm.configure_optimizers()["optimizer"]
    sharded = sharded and hasattr(opt, "step_shards")

    micro = [0]
    use_graph = args.hip_graph == "on" or (args.hip_graph == "auto" and getattr(lm.model, "prefers_hip_graph", False))
    graphed = [None]

    def step(i):
        if graphed[0] is not None:
            loss = graphed[0](make_batch(case))
        else:
            loss = lm.training_step(make_batch(case), i)
            if (micro[0] + 1) % args.accumulate == 0:
                ddp.arm()   # (N > 1, several buckets: the exchange starts inside this backward)
            (loss / args.accumulate if args.accumulate > 1 else loss).backward()
        micro[0] += 1
        if micro[0] % args.accumulate == 0:   # non-stepping micro-batches neither sync nor step (trainer.yaml:58)
            ddp.all_reduce_grads()
            if sharded:
                opt.step_shards(ddp.shards())
                ddp.all_gather_params()
            else:
                opt.step()
            ddp.zero_grad()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.setup_steps):   # lazy one-time initialisation, not part of the W + K contract
        step(-1 - i)
    barrier()
    probe = None
    graph_single_stream = False
    if args.hip_graph == "auto" and not use_graph and hasattr(lm.model, "native_rollout"):
        # Launch-mode probe (also outside the W + K contract).  The native HalfUNet step is ~230 launches: issued eagerly they cost
        # the host ~2 ms, less than the GPU needs, and eager launching is then faster than replaying the captured step -- but on a
        # slow or busy host (eight ranks on one node) the step becomes host-bound, and a replay (one launch) is the faster way.
        # Three modes are measured: eager, the captured two-stream step, and the captured step with the weight gradients on the
        # main stream (the two-stream capture is dealt over four hardware queues and loses its overlap; the one-stream replay costs
        # what eager costs on the GPU and nothing on the host: profiles/r04_graph_vs_eager.txt).  With several ranks every rank
        # measures, the MAXIMA over the ranks decide, so all ranks pick the same mode (captures hold no collective: the gradient
        # exchange stays outside the graph).
        def per_step(n):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for i in range(n):
                step(-100 - i)
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n

        from py4cast_amd.trainer import GraphedTrainingStep

        def graph_time(single):
            L.lib().p4c_side_stream_enable(0 if single else 1)
            t, ok = float("inf"), True
            try:
                ddp.zero_grad()
                graphed[0] = GraphedTrainingStep(lm, make_batch(case), loss_scale=1.0 / args.accumulate)
            except Exception as exc:  # noqa: BLE001  (a step that cannot be captured simply stays eager)
                print(f"bench: HIP-graph probe failed ({type(exc).__name__}: {exc}); eager launching", file=sys.stderr)
                graphed[0], ok = None, False
            ddp.zero_grad()
            per_step(1)            # (run on every rank whether or not its capture worked: the ranks' collective counts stay equal)
            t_run = per_step(3)
            graphed[0] = None
            ddp.zero_grad()
            L.lib().p4c_side_stream_enable(1)
            return t_run if ok else t

        times = [per_step(3), graph_time(False), graph_time(True)]
        if world > 1:
            tt = torch.tensor([min(t, 1e9) for t in times], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            times = [float(v) if float(v) < 1e9 else float("inf") for v in tt.tolist()]
        t_eager, t_graph, t_graph1 = times
        best_graph = min(t_graph, t_graph1)
        use_graph = best_graph < 0.97 * t_eager
        graph_single_stream = use_graph and t_graph1 <= t_graph
        ms = lambda t: None if t == float("inf") else 1e3 * t      # noqa: E731
        probe = {"eager_ms_per_step": ms(t_eager), "graph_ms_per_step": ms(t_graph), "graph_single_stream_ms_per_step": ms(t_graph1),
                 "agreed_over_ranks": world, "chosen": ("graph, single stream" if graph_single_stream else "graph") if use_graph else "eager"}
        barrier()
    # The timed region is ~0.1 s: a generational garbage collection of this process (tens of thousands of tracked objects once
    # torch and the model are loaded) is a 10-50 ms host pause that starves the GPU.  The collector is parked from here on --
    # BEFORE the warm-up steps, so that the caching allocator reaches its steady state under the same object lifetimes as the
    # timed steps (parked only for the timed steps, the second of them grew the pool by one device allocation).
    import gc

    gc.collect()
    gc.freeze()
    gc_was_enabled = gc.isenabled()
    gc.disable()
    for i in range(args.warmup):
        step(i)
    timed = getattr(lm.model, "timed_entry_points", None) or (
        "p4c_build_x", "p4c_ar_update_fwd", "p4c_weighted_loss_fwd", "p4c_weighted_loss_bwd", "p4c_ar_update_bwd")
    has_roofline = hasattr(lm.model, "roofline")
    if args.kernel_times or not has_roofline or getattr(lm.model, "roofline_from_entry_points", False):
        L.enable_kernel_timing(timed)
    if args.kernel_times:
        L.lib().p4c_prof_enable(7, 4096)
    elif has_roofline and not getattr(lm.model, "roofline_from_entry_points", False):
        # roofline leg: only the dominant kernel's forward-plan launches at full resolution get event markers, and only those of
        # the first four timed steps (3 per AR step): an event record costs the stream a bubble of several microseconds, so
        # marking all of them (90 launches at the defaults) lowered the measured throughput by ~2 %
        L.lib().p4c_prof_enable(1, min(args.steps, 4) * T * 3)
        L.lib().p4c_prof_filter(B * H * W)
    barrier()
    roof_model = None
    graph_note = None
    work = None
    if use_graph:
        # kernel timings for the roofline object come from two eager steps (a replayed graph has no per-call host hooks);
        # then the micro-batch is captured and the warm-up + timed steps replay it
        from py4cast_amd.trainer import GraphedTrainingStep

        L.WORK[0] = {"bytes": 0.0, "flops": 0.0, "calls": 0, "unstated": 0}
        for i in range(2):
            step(args.warmup + i)
        work, L.WORK[0] = L.WORK[0], None
        work = {k: v / 2 for k, v in work.items()}        # per step
        ktimes = L.kernel_times()
        roof_model = lm.model.roofline(ktimes, B=B, H=H, W=W) if (rank == 0 and hasattr(lm.model, "roofline")) else None
        L.enable_kernel_timing(None)
        L.lib().p4c_prof_enable(0, 0)   # no event markers inside a capture
        ddp.zero_grad()
        if graph_single_stream:
            L.lib().p4c_side_stream_enable(0)
        try:
            graphed[0] = GraphedTrainingStep(lm, make_batch(case), loss_scale=1.0 / args.accumulate)
            graph_note = graphed[0].verified
        except Exception as exc:  # noqa: BLE001  (not capturable, or a replay that does not reproduce the eager step: stay eager)
            print(f"bench: HIP graph not used ({type(exc).__name__}: {exc}); eager launching", file=sys.stderr)
            graphed[0], use_graph, graph_note = None, False, f"rejected: {type(exc).__name__}"
        ddp.zero_grad()
        for i in range(args.warmup):
            step(i)
        barrier()
    # one event per step boundary (a record costs the stream a few microseconds against a ~5 ms step): min / median / max
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    # (collector parked since before the warm-up steps; device allocations inside the timed region are counted)
    allocs_before = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    trace_allocs = os.environ.get("P4C_BENCH_TRACE_ALLOCS") == "1"   # debugging: which timed step allocates device memory
    alloc_steps = []
    for i in range(args.steps):
        loss = step(args.warmup + i).detach()   # (holding the loss itself would keep the step's autograd graph -- its state buffers -- alive into the next step)
        marks[i + 1].record()
        if trace_allocs:
            n = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
            if n != allocs_before + len(alloc_steps):
                alloc_steps.append(i)
    # wall time of the enqueue loop per step: with a full launch queue it tracks the GPU's step time (back-pressure); the host's own
    # cost of a step, measured with an empty queue, is ~1.9 ms (tools/diagnostics/host_time.py)
    host_enqueue_ms = (time.perf_counter() - t0) / args.steps * 1e3
    barrier()
    dt = time.perf_counter() - t0
    if trace_allocs:
        print(f"bench: device allocations in timed steps {alloc_steps}; reserved {torch.cuda.memory_reserved(device) / 2**20:.0f} MiB", file=sys.stderr)
    device_allocs = torch.cuda.memory_stats(device).get("num_device_alloc", 0) - allocs_before
    if gc_was_enabled:
        gc.enable()
    gc.unfreeze()
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    if not use_graph:
        ktimes = L.kernel_times()
        roof_model = lm.model.roofline(ktimes, B=B, H=H, W=W) if (rank == 0 and hasattr(lm.model, "roofline")) else None
        L.enable_kernel_timing(None)
    L.lib().p4c_prof_enable(0, 0)
    extra = None
    if has_roofline and not use_graph and not getattr(lm.model, "roofline_from_entry_points", False) \
            and hasattr(lm.model, "launch_times"):   # (every rank: the step contains the gradient exchange)
        # ONE extra un-timed step with every tagged launch bracketed by events: data-gradient and weight-gradient launch times
        # of the roofline kernel's siblings (they overlap each other in the backward plan, hence reported apart)
        L.lib().p4c_prof_enable(7, 4096)
        step(args.warmup + args.steps)
        torch.cuda.synchronize()
        extra = lm.model.launch_times(B=B, H=H, W=W) if rank == 0 else None
        L.lib().p4c_prof_enable(0, 0)
    affinity_all = None
    share = None
    if rank == 0 and world == 1 and not args.no_native_share:
        saved_graph, graphed[0] = graphed[0], None          # one EAGER step under the profiler (a replay has the same kernels)
        if work is None:
            L.WORK[0] = {"bytes": 0.0, "flops": 0.0, "calls": 0, "unstated": 0}
        try:
            share = native_share_of_one_step(lambda: step(args.warmup + args.steps + 1))
        except Exception as exc:  # noqa: BLE001  (no device tracing in this environment: the field stays null)
            print(f"bench: native share not measured ({type(exc).__name__}: {exc})", file=sys.stderr)
        if work is None:
            work, L.WORK[0] = L.WORK[0], None
        graphed[0] = saved_graph
    if world > 1:
        tmax = torch.tensor([dt, host_enqueue_ms], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt, host_enqueue_ms = float(tmax[0].item()), float(tmax[1].item())      # (host loop: the slowest rank's)
        affinity_all = [None] * world
        torch.distributed.all_gather_object(affinity_all, affinity)

    if rank == 0:
        N = H * W
        c_in = F + Fs + Ff
        roof = None
        if roof_model is not None:
            roof = roof_model
            if extra:
                roof.update(extra)
            if hasattr(lm.model, "step_algorithmic_bytes"):
                # the whole step against the HBM roofline: every launch's compulsory reads + writes (per-family table in DESIGN.md
                # section 6, computed by HalfUNetMI355X.step_algorithmic_bytes) over the measured step time
                tot, table = lm.model.step_algorithmic_bytes(B, H, W, F, Fs, Ff, T)
                step_s = dt / args.steps
                roof["step"] = {"bound": "hbm", "algorithmic_bytes": tot, "achieved": tot / step_s / 1e9, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": tot / step_s / 1e9 / HBM_PEAK_GBS, "ms_per_step": step_s * 1e3,
                                "floor_ms_at_peak": tot / (HBM_PEAK_GBS * 1e9) * 1e3,
                                "bytes_by_family_mb": {k: round(v / 1e6, 1) for k, v in table.items()}}
                # measured HBM bytes of a whole step, when a committed PMC run of THIS configuration exists (separate rocprofv3 --pmc
                # FETCH_SIZE / WRITE_SIZE passes over every dispatch, tools/diagnostics/r04_pmc_step.sh): only for the default shape
                if (B, H, W, F, Fs, Ff, T) == (2, 512, 512, 60, 4, 5, 3) and args.dtype == "bf16":
                    hb, src = L.committed_traffic(("r06_pmc_traffic_step.json", "r05_pmc_traffic_step.json", "r04_pmc_traffic_step.json"), ("hbm_bytes_per_step", "total"))
                    roof["step"]["traffic"] = hb
                    roof["step"]["traffic_source"] = src
            wg = (extra or {}).get("wgrad_avg_launch_ms_overlapped")
            if wg and roof.get("bound") == "hbm":
                # the weight-gradient kernel of the same layers, in the step (it runs beside the backward chain on the side stream):
                # algorithmic bytes = its two operand maps (x, dY) -- the figure the forward kernel is priced with
                ab = roof.get("algorithmic_bytes_per_launch")
                roof["wgrad_in_step"] = {"kernel": "weight gradient of the 3x3 conv 64->64 at full resolution (overlapped with the backward chain)",
                                         "algorithmic_bytes_per_launch": ab, "avg_launch_ms": wg,
                                         "achieved": ab / (wg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": ab / (wg * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if roof is not None and "step" not in roof and work and work["bytes"] > 0:
            # the widened models: the whole step against both roofs from the algorithmic bytes / matrix flops that the wrappers of
            # the native entry points state call by call (py4cast_amd/_lib.py::WORK; library calls state nothing: `unstated_calls`)
            step_s = dt / args.steps
            roof["step"] = {"bound": "hbm", "algorithmic_bytes": work["bytes"], "achieved": work["bytes"] / step_s / 1e9,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": work["bytes"] / step_s / 1e9 / HBM_PEAK_GBS,
                            "ms_per_step": step_s * 1e3, "floor_ms_at_peak": work["bytes"] / (HBM_PEAK_GBS * 1e9) * 1e3,
                            "matrix_flops": work["flops"], "matrix_tflops": work["flops"] / step_s / 1e12,
                            "matrix_frac_of_bf16_peak": work["flops"] / step_s / 1e12 / 2500.0,
                            "native_calls": int(work["calls"]), "unstated_calls": int(work["unstated"]),
                            "note": "bytes / flops as stated by the wrappers of the native entry points, one eager step"}
        if roof is None and ktimes and not hasattr(lm.model, "roofline"):
            # HBM-bound rollout kernels: algorithmic bytes per launch
            alg = {
                "p4c_build_x": 4.0 * B * N * (2 * c_in),
                "p4c_ar_update_fwd": 4.0 * B * N * (4 * F + 2),
                "p4c_weighted_loss_fwd": 4.0 * B * T * N * (2 * F),
                "p4c_weighted_loss_bwd": 4.0 * B * T * N * (3 * F),
                "p4c_ar_update_bwd": 4.0 * B * N * (3 * F),
            }
            name = max(ktimes, key=lambda k: ktimes[k][0] * ktimes[k][1])
            gbs = alg[name] / (ktimes[name][1] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": name, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": ktimes[name][1],
                    "launches": ktimes[name][0]}
        out = {
            "metric": "AR-training samples/sec (512x512x60 grid, 3-step rollout)" if (H, W, F, T) == (512, 512, 60, 3)
                      else f"AR-training samples/sec ({H}x{W}x{F} grid, {T}-step rollout)",
            "value": world * B * args.steps / dt,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "step_ms": {"min": step_ms[0], "median": step_ms[len(step_ms) // 2], "max": step_ms[-1],
                        "note": "per-step durations between HIP events on the compute stream of rank 0 inside the timed region"},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype if args.model != "Identity" else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.model} {args.strategy} rollout T={T}, grid {H}x{W}x{F} (+{Ff} forcings, {Fs} statics), "
                            f"WeightedLoss(MSE), AdamW, B={B}/GPU"
                            + ({"published": " [UNETR++ block as published: x_SA merge, conv8 = Sequential(Dropout2d(0.1), Conv) drawn every step, E = F]",
                                "published-nodrop": " [UNETR++ block as published, conv8 dropout p = 0]",
                                "restated": " [restated UNETR++ block of rounds 2-5: not checkpoint compatible with mfai -- py4cast_amd/unetrpp.py]"}
                               [args.unetrpp_block] if args.model.lower().startswith("unetrpp") else ""),
                "global_batch": world * B,
                "gradient_exchange": None if world == 1 else {
                    "mode": "reduce-scatter + sharded AdamW + all-gather" if sharded else "all-reduce",
                    "buckets": len(ddp.buckets), "bytes": n_grad_bytes, "overlapped_with_backward": bool(ddp.overlap),
                    "buckets_issued_inside_last_backward": int(ddp.issued_in_backward)},
                "parallelism": f"dp{world}" + (" (ranks sharing one GPU over gloo: functional test, not a scaling number)" if share_gpu else ""),
                "border_size": args.border,
                "setup_steps": args.setup_steps,
                "accumulate_grad_batches": args.accumulate,
                "hip_graph": bool(use_graph), "hip_graph_check": graph_note,
                "deterministic_library_solvers": bool(torch.backends.cudnn.deterministic),
                "peak_hbm_gib": round(torch.cuda.max_memory_allocated(device) / 2**30, 2),
                "device_allocs_in_timed_region": int(device_allocs),
                "host_loop_ms_per_step": round(host_enqueue_ms, 3),
                "host_cores_per_rank": affinity_all if world > 1 else affinity,
                "host_affinity_policy": __import__("py4cast_amd.trainer", fromlist=["AFFINITY_POLICY"]).AFFINITY_POLICY[0],
                "environment_settings": L.environment_settings(),
                "launch_mode_probe": probe,
            },
            "loss": float(loss.detach()),
            "native_share": None if share is None else {"of_gpu_kernel_time": round(share[0], 4), "ms_by_origin": share[1], "kernels": share[2],
                                                        "note": "one eager step under torch.profiler: kernels named p4c:: are this repository's"},
            "roofline": roof,
            "kernel_ms": {k: {"calls": v[0], "avg_ms": round(v[1], 4)} for k, v in ktimes.items()},
        }
        if (world == 1 and not args.no_fp32_flavour and args.dtype == "bf16" and args.model == "HalfUNet"
                and hasattr(lm.model, "native_rollout")):
            # the parity flavour (exact fp32 matrix cores, fp32 storage; <= 1e-4 vs the oracle) of the SAME workload, measured by
            # the same process after the headline run: a short run of its own (2 setup + 2 warm-up + 6 timed steps)
            del lm, ddp, opt
            torch.cuda.empty_cache()
            out["fp32_flavour"] = fp32_flavour(args, case, info, device)
            if not args.no_larger_batch and args.batch < args.larger_batch:
                # the SAME workload with a per-GPU batch sized for this GPU's memory rather than the reference yaml's 2
                # (config/CLI/dataset/titan.yaml:7; 15 GiB of 288 at B=8): `value` above stays the B=2 configuration of BASELINE.md
                torch.cuda.empty_cache()
                big = synthetic_case(1234 + rank, args.larger_batch, T, 1, H, W, F, Ff, Fs, args.border, device)
                out["larger_batch"] = fp32_flavour(args, big, info, device, steps=8, warmup=4, dtype="bf16", batch=args.larger_batch)
                out["larger_batch"]["note"] = ("same rollout / loss / optimizer and model at a per-GPU batch the 288 GB of HBM invite; not the "
                                               "headline `value` (BASELINE.md fixes B=2 per GPU)")
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        if others is not None:
            # BASELINE configurations 3 / 4 / 5, each in a fresh process (this one's GPU work is done; its memory is released first)
            try:
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                others.stdin.write("go\n")
                others.stdin.flush()
                line = others.stdout.readline()
                out["other_configs"] = json.loads(line) if line.strip() else {"error": "the launcher returned nothing"}
                out["other_configs"]["note"] = ("bench.py --model X --steps 5 --warmup 2 per configuration (HIP-graph replay where the model "
                                                "prefers it), each in its own process after the headline legs; not part of `value`")
            except Exception as exc:  # noqa: BLE001
                out["other_configs"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
