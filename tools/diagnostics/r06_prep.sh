#!/bin/bash
# round 6: batched weight-image preparation -- tests, then the UNETR++ / Swin step A/B (batched / one launch per weight)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06y}; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gemm_gpu.py tests/test_unetrpp_gpu.py tests/test_widen_gpu.py tests/test_swin_golden_gpu.py tests/test_bench_size_gpu.py -q 2>&1 | tail -25 > $O/tests.txt
tail -5 $O/tests.txt
for rep in 1 2; do
  timeout 600 python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/unetrpp_batch_$rep.json 2> $O/unetrpp_batch_$rep.err
  P4C_R06_NO_BATCH_PREP=1 timeout 600 python3 tools/diagnostics/r06_gnn_ab.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/unetrpp_single_$rep.json 2> $O/unetrpp_single_$rep.err
  timeout 600 python3 bench.py --model SwinUNetR --no-cpu-baseline --no-other-configs > $O/swin_batch_$rep.json 2> $O/swin_batch_$rep.err
  P4C_R06_NO_BATCH_PREP=1 timeout 600 python3 tools/diagnostics/r06_gnn_ab.py --model SwinUNetR --no-cpu-baseline --no-other-configs > $O/swin_single_$rep.json 2> $O/swin_single_$rep.err
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, round(d['ms_per_step'],2), d['config'].get('hip_graph_check'))
    except Exception as e:
        print(f, 'failed', e)
PY
