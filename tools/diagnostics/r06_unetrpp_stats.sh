#!/bin/bash
# round 6: per-kernel statistics of the eager UNETR++ step (same command as r06_profile.sh:61) for the current tree
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06n}
mkdir -p $O
rm -rf /tmp/pu
rocprofv3 --kernel-trace --stats -d /tmp/pu -- python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 3 --warmup 1 --no-cpu-baseline --no-native-share --no-other-configs --hip-graph off > /dev/null 2>&1
python3 tools/diagnostics/rocpd_stats.py /tmp/pu/*/*_results.db $O/unetrpp_bf16_kernel_stats.csv
head -3 $O/unetrpp_bf16_kernel_stats.csv | cut -c1-200
