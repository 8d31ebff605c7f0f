#!/bin/bash
# round 6: launch-geometry knobs of the tall-skinny kernels inside the UNETR++ step (diagnostic library: P4C_TS_APPLY_WGS = workgroups per CU
# the apply grid is capped at, default 4; P4C_TS_SPLIT_TOKENS = tokens per gram split from 4 096 tokens on, default 256)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06z}; mkdir -p $O
B="--model UNetRPP --strategy diff_ar --pred-steps 6 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs"
timeout 600 python3 tools/diagnostics/bench_diag.py $B > $O/default_1.json 2>/dev/null
for w in 2 8 16; do P4C_TS_APPLY_WGS=$w timeout 600 python3 tools/diagnostics/bench_diag.py $B > $O/apply_wgs_$w.json 2>/dev/null; done
for t in 128 512 1024; do P4C_TS_SPLIT_TOKENS=$t timeout 600 python3 tools/diagnostics/bench_diag.py $B > $O/split_tokens_$t.json 2>/dev/null; done
timeout 600 python3 tools/diagnostics/bench_diag.py $B > $O/default_2.json 2>/dev/null
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, round(d['ms_per_step'],2))
    except Exception as e:
        print(f, 'failed', e)
PY
