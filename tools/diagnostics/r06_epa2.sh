#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06o}
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_unetrpp_gpu.py -x -q 2>&1 | tail -25 > $O/tests.txt
tail -4 $O/tests.txt
bash tools/diagnostics/r06_unetrpp_stats.sh ${1:-r06o} > /dev/null 2>&1
timeout 600 python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench_native.json 2> $O/bench_native.err
python3 - <<PY
import json
d=json.loads([l for l in open('$O/bench_native.json') if l.startswith('{')][-1])
print('native', d['ms_per_step'], d['config'].get('native_kernel_share'))
PY
