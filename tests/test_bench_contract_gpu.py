"""The driver's contract for bench.py on one GPU: ONE JSON line on stdout with the metric of BASELINE.json, whole-job throughput,
the roofline object of the dominant kernel and the CPU baseline (oracle, bounded sample) -- checked on a small grid so that the
test takes seconds; the numbers themselves are the benchmark's business."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_bench_prints_one_json_line_with_the_contract_fields(gpu_device):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--grid", "128", "128",
                          "--cpu-seconds", "1", "--no-fp32-flavour", "--no-larger-batch", "--hip-graph", "off", "--no-other-configs"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "samples/s" and d["dtype"] == "bf16" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - d["config"]["global_batch"] * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    sm = d["step_ms"]
    assert sm["min"] <= sm["median"] <= sm["max"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"]
    assert d["config"]["device_allocs_in_timed_region"] == 0


def test_default_run_reports_the_other_baseline_configurations(gpu_device):
    """The default one-GPU run also times BASELINE configurations 3 / 4 / 5 (SwinUNetR, HiLAM, UNetRPP 6-step diff_ar), each in a fresh
    child process started by a launcher that itself was started before the parent's first GPU call: the ONE JSON line carries
    `other_configs` = {name: {ms_per_step, native_share, roofline_step_frac, workload}} (here on a small grid and a narrow UNETR++)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--grid", "128", "128", "--hidden", "512",
                          "--no-cpu-baseline", "--no-fp32-flavour", "--no-larger-batch", "--hip-graph", "off"],
                         capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    oc = d["other_configs"]
    assert set(oc) >= {"SwinUNetR", "HiLAM", "UNetRPP"}, oc
    for name in ("SwinUNetR", "HiLAM", "UNetRPP"):
        assert "error" not in oc[name], oc[name]
        assert oc[name]["ms_per_step"] > 0 and oc[name]["steps"] == 5 and name in oc[name]["workload"]
        assert oc[name]["native_share"] is None or 0 < oc[name]["native_share"] <= 1
    assert "diff_ar rollout T=6" in oc["UNetRPP"]["workload"] and "as published" in oc["UNetRPP"]["workload"]
