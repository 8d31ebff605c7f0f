#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06p}
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_ts_reduce_gpu.py tests/test_unetrpp_gpu.py tests/test_widen_gpu.py tests/test_gemm_gpu.py -x -q 2>&1 | tail -25 > $O/tests.txt
tail -6 $O/tests.txt
for rep in 1 2; do
timeout 600 python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench_native_$rep.json 2> $O/bench_native_$rep.err
P4C_UNETRPP_LIB_DROPOUT=1 timeout 600 python3 tools/diagnostics/bench_diag.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench_libdrop_$rep.json 2> $O/bench_libdrop_$rep.err
done
P4C_EPA_LIB_MERGE=1 P4C_EPA_LIB_PROJ=1 timeout 600 python3 tools/diagnostics/bench_diag.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench_libboth_1.json 2> $O/bench_libboth_1.err
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, round(d['ms_per_step'],2), d['config'].get('native_kernel_share'))
    except Exception as e:
        print(f, 'failed', e)
PY
