#!/bin/bash
# round 6, call 5: re-run of the two failed tests; gemm_nt XCD-aware tile order A/B (micro + UNETR++ step + Swin step); in-kernel clock and
# prologue / steady / drain timeline of conv3x3_bf16_rows for the full kernel, the compute-only and the memory-only builds (512^2 and 256^2)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06e; mkdir -p $O
timeout 900 python3 -m pytest tests/test_unetrpp_gpu.py tests/test_bench_contract_gpu.py -x -q -k "build_x or other_baseline or published" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 900 python3 -m pytest tests/test_gemm_gpu.py -x -q > $O/test_gemm.txt 2>&1; tail -3 $O/test_gemm.txt
echo "== gemm micro, XCD-aware order (default)"; python3 tools/diagnostics/gemm_micro.py 2>&1 | tee $O/gemm_micro_xcd.txt
echo "== gemm micro, tile = blockIdx.x (round 5)"; P4C_LIB_PATH=tools/diagnostics/libs/lib_gemm_noxcd.so python3 tools/diagnostics/gemm_micro.py 2>&1 | tee $O/gemm_micro_noxcd.txt
U="--model UNetRPP --strategy diff_ar --pred-steps 6 --steps 5 --warmup 2 --no-cpu-baseline --no-native-share --unetrpp-block restated"
python3 bench.py $U > $O/unetrpp_xcd.json 2>/dev/null
P4C_LIB_PATH=tools/diagnostics/libs/lib_gemm_noxcd.so python3 bench.py $U > $O/unetrpp_noxcd.json 2>/dev/null
python3 bench.py $U > $O/unetrpp_xcd2.json 2>/dev/null
S="--model SwinUNetR --steps 8 --warmup 3 --no-cpu-baseline --no-native-share"
python3 bench.py $S > $O/swin_xcd.json 2>/dev/null
P4C_LIB_PATH=tools/diagnostics/libs/lib_gemm_noxcd.so python3 bench.py $S > $O/swin_noxcd.json 2>/dev/null
for f in $O/*.json; do echo $f $(python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'])"); done
for lib in stamps stamps_exp6 stamps_exp16; do
  for shape in 2x512x512 2x256x256; do
    for mode in plain ts; do
      echo "== $lib $shape $mode"
      WARM=50000 P4C_LIB_PATH=tools/diagnostics/libs/lib_rows_$lib.so python3 tools/diagnostics/rows_stamps.py $shape $mode 2>&1 | head -70
    done
  done
done > $O/rows_stamps.txt 2>&1
grep -E "^==|clock|loader:|compute:" $O/rows_stamps.txt
