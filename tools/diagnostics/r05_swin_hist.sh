#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace -d /tmp/sh -- python3 bench.py --model SwinUNetR --steps 3 --warmup 1 --no-cpu-baseline --no-native-share --hip-graph off > /dev/null 2>&1
db=$(find /tmp/sh -name "*.db" | head -1)
for k in window_attn_bwd window_attn_fwd dbias_reduce inorm::reduce inorm::apply finalize_bwd finalize_fwd gemm_nt_kernel row_layernorm_bwd; do python3 tools/diagnostics/kernel_hist.py $db $k | head -9; done
