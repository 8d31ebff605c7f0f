#!/bin/bash
# UNETR++: layer scale gamma folded into the output projections' weight images by the preparation kernel (default) against the torch
# products gamma * W, gamma * b (P4C_UNETRPP_GAMMA_MUL=1): tests, then the bench line of both
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 -m pytest tests/test_gemm_gpu.py tests/test_unetrpp_gpu.py -m gpu -x -q 2>&1 | tail -6
for v in 1 0; do
  if [ $v = 1 ]; then export P4C_UNETRPP_GAMMA_MUL=1; else unset P4C_UNETRPP_GAMMA_MUL; fi
  python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('torch products' if $v else 'folded', d['ms_per_step'], d['native_share']['of_gpu_kernel_time'], d['native_share']['kernels'], d['loss'])"
done
