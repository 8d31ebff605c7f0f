#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace -d /tmp/hh -- python3 bench.py --model HiLAM --steps 3 --warmup 1 --no-cpu-baseline --no-native-share --hip-graph off > /dev/null 2>&1
db=$(find /tmp/hh -name "*.db" | head -1)
python3 tools/diagnostics/kernel_hist.py $db row_mlp_bwd | head -30
python3 tools/diagnostics/kernel_hist.py $db row_mlp_fwd | head -12
python3 tools/diagnostics/kernel_hist.py $db row_gemm_kernel | head -12
