#!/bin/bash
# round 6, call 1: the new mesh-GNN tests, the GNN part of the suite, same-box A/B of HiLAM (old routes / deferred reduce only / all)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06a; mkdir -p $O
timeout 900 python3 -m pytest tests/test_nodeproj_gpu.py -x -q > $O/test_nodeproj.txt 2>&1
tail -5 $O/test_nodeproj.txt
timeout 1500 python3 -m pytest tests/test_widen_gpu.py tests/test_bench_size_gpu.py -x -q -k "graphlam or hilam or mesh or row_mlp or row_linear or graphed or trainer or widened" > $O/test_gnn.txt 2>&1
tail -5 $O/test_gnn.txt
B="--steps 10 --warmup 3 --no-cpu-baseline --no-native-share"
for m in HiLAM; do
  P4C_R06_OLD_PROJ=1 P4C_R06_NO_DEFER=1 python3 tools/diagnostics/r06_gnn_ab.py --model $m $B > $O/${m}_old.json 2> $O/${m}_old.err
  P4C_R06_OLD_PROJ=1 python3 tools/diagnostics/r06_gnn_ab.py --model $m $B > $O/${m}_defer_only.json 2> $O/${m}_defer_only.err
  python3 bench.py --model $m $B > $O/${m}_new.json 2> $O/${m}_new.err
  P4C_R06_OLD_PROJ=1 P4C_R06_NO_DEFER=1 python3 tools/diagnostics/r06_gnn_ab.py --model $m $B > $O/${m}_old2.json 2> $O/${m}_old2.err
  python3 bench.py --model $m $B > $O/${m}_new2.json 2> $O/${m}_new2.err
done
for f in $O/*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('config',{}).get('launch_mode'))
except Exception as e: print('ERR', e)
"; done
