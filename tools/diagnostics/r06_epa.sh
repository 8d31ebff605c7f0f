#!/bin/bash
# round 6: EPA token-axis projection on the tall-skinny kernels -- parity tests, then the UNETR++ step A/B (native vs library projection)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r06m
timeout 1200 python3 -m pytest tests/test_unetrpp_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r06m/tests.txt
tail -6 gpurun_out/r06m/tests.txt
for rep in 1 2; do
  timeout 600 python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/r06m/bench_native_$rep.json 2> gpurun_out/r06m/bench_native_$rep.err
  P4C_EPA_LIB_PROJ=1 timeout 600 python3 tools/diagnostics/bench_diag.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/r06m/bench_lib_$rep.json 2> gpurun_out/r06m/bench_lib_$rep.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06m/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d['ms_per_step'], d['config'].get('native_kernel_share'), d['config'].get('hip_graph'))
    except Exception as e:
        print(f, 'failed', e)
PY
