#!/bin/bash
# round 6, call 7: EPA token projection on the tall-skinny kernels (tests + UNETR++ step A/B through the diagnostic switch)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06g; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_unetrpp_gpu.py -x -q > $O/test_unetrpp.txt 2>&1; tail -5 $O/test_unetrpp.txt
timeout 1200 python3 -m pytest tests/test_widen_gpu.py tests/test_gemm_gpu.py -x -q -k "swin or compact or batch_norm or layer_norm or gemm" > $O/test_misc.txt 2>&1; tail -3 $O/test_misc.txt
U="--model UNetRPP --strategy diff_ar --pred-steps 6 --steps 5 --warmup 2 --no-cpu-baseline --no-native-share --unetrpp-block restated"
python3 bench.py $U > $O/unetrpp_native_proj.json 2>$O/a.err
P4C_EPA_LIBRARY_PROJ=1 python3 tools/diagnostics/bench_diag.py $U > $O/unetrpp_library_proj_diaglib.json 2>$O/b.err
python3 tools/diagnostics/bench_diag.py $U > $O/unetrpp_native_proj_diaglib.json 2>$O/c.err
python3 bench.py $U > $O/unetrpp_native_proj2.json 2>/dev/null
for f in $O/*.json; do echo $f $(python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['hip_graph'])"); done
tail -2 $O/*.err
