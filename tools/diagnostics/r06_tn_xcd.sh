#!/bin/bash
# (the kernel-side experiment is NOT in the tree: tools/diagnostics/patches/r06_tn_xcd.patch; apply it, build, then gemm_build.sh tn_noxcd -DP4C_TN_XCD=0)
# round 6: gemm_tn, the nine tap tiles of a unit on one XCD (default) against tile = blockIdx.x (lib_gemm_tn_noxcd.so, -DP4C_TN_XCD=0)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06zb}; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gemm_gpu.py -q 2>&1 | tail -4 > $O/tests.txt; tail -2 $O/tests.txt
echo "== gemm micro, XCD-grouped taps"; python3 tools/diagnostics/gemm_micro.py 2>&1 | grep -E "conv3x3" | tee $O/gemm_micro_xcd.txt
echo "== gemm micro, tile = blockIdx.x"; P4C_LIB_PATH=tools/diagnostics/libs/lib_gemm_tn_noxcd.so python3 tools/diagnostics/gemm_micro.py 2>&1 | grep -E "conv3x3" | tee $O/gemm_micro_noxcd.txt
U="--model UNetRPP --strategy diff_ar --pred-steps 6 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs"
S="--model SwinUNetR --no-cpu-baseline --no-other-configs"
for rep in 1 2; do
  python3 bench.py $U > $O/unetrpp_xcd_$rep.json 2>/dev/null
  P4C_LIB_PATH=tools/diagnostics/libs/lib_gemm_tn_noxcd.so python3 bench.py $U > $O/unetrpp_noxcd_$rep.json 2>/dev/null
  python3 bench.py $S > $O/swin_xcd_$rep.json 2>/dev/null
  P4C_LIB_PATH=tools/diagnostics/libs/lib_gemm_tn_noxcd.so python3 bench.py $S > $O/swin_noxcd_$rep.json 2>/dev/null
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, round(d['ms_per_step'],2))
    except Exception as e:
        print(f, 'failed', e)
PY
