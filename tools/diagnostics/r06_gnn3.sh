#!/bin/bash
# round 6, call 3: tests of the prologue / per-wave partial / fold changes + HiLAM, HiLAMParallel, GraphLam lines + HiLAM histogram
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06c; mkdir -p $O
timeout 900 python3 -m pytest tests/test_nodeproj_gpu.py -x -q > $O/test_nodeproj.txt 2>&1
tail -4 $O/test_nodeproj.txt
timeout 1500 python3 -m pytest tests/test_widen_gpu.py tests/test_bench_size_gpu.py -x -q -k "graphlam or hilam or mesh or row_mlp or row_linear or graphed or trainer or widened" > $O/test_gnn.txt 2>&1
tail -4 $O/test_gnn.txt
B="--steps 10 --warmup 3 --no-cpu-baseline"
for m in HiLAM HiLAMParallel GraphLam; do
  python3 bench.py --model $m $B > $O/${m}.json 2> $O/${m}.err
done
for f in $O/*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('native_share'), (d.get('roofline') or {}).get('frac'))
except Exception as e: print('ERR', e)
"; done
rm -rf /tmp/hh; rocprofv3 --kernel-trace -d /tmp/hh -- python3 bench.py --model HiLAM --steps 3 --warmup 1 --no-cpu-baseline --no-native-share --hip-graph off > /dev/null 2>&1
db=$(find /tmp/hh -name "*.db" | head -1)
for k in row_mlp_bwd row_mlp_fwd; do python3 tools/diagnostics/kernel_hist.py $db $k | head -10; done > $O/hilam_hist.txt 2>&1
cat $O/hilam_hist.txt
