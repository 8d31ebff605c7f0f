#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 -m pytest tests/test_unetrpp_gpu.py -m gpu -x -q 2>&1 | tail -6
for i in 1 2; do
  python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['native_share']['of_gpu_kernel_time'], d['native_share']['kernels'], d['loss'])"
done
