#!/bin/bash
# Round-4 final evidence run (one gpurun call): full GPU suite + smoke, default bench line, Titan-shape line, 500-step run, rocprofv3
# kernel trace of the bench command (default and Titan shape).  Outputs under gpurun_out/r04final/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r04final; mkdir -p $O
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --grid 512 640 --features 21 --forcings 21 --border 10 > $O/bench_titan.json 2> $O/bench_titan.err
python3 bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-fp32-flavour --no-larger-batch > $O/bench_500_steps.json 2>/dev/null
tools/diagnostics/r04_trace.sh $O/trace > /dev/null 2>&1
tools/diagnostics/r04_trace.sh $O/trace_titan --grid 512 640 --features 21 --forcings 21 --border 10 > /dev/null 2>&1
cat $O/gpu_tests.txt $O/smoke.txt | tail -8
python3 - <<'PY'
import json
for f in ("bench_default", "bench_titan", "bench_500_steps"):
    try:
        d = json.loads(open(f"gpurun_out/r04final/{f}.json").readlines()[-1])
        print(f, round(d["value"], 1), round(d["ms_per_step"], 3), d["step_ms"], d["roofline"].get("frac"), (d["roofline"].get("step") or {}).get("frac"))
    except Exception as e:
        print(f, "FAILED", e)
PY
