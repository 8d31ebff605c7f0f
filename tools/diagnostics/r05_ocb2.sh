#!/bin/bash
# out_conv_bwd at 2 / 3 / 4 workgroups per CU (P4C_OCB_PER_CU, diagnostic library): what the AR-step-backward fusion (32 more prefetch
# registers -> two workgroups per CU) would start from.  Per-kernel durations of the HalfUNet bench under rocprofv3.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05ocb; mkdir -p $O
for r in 2 3 4; do
  export P4C_OCB_PER_CU=$r
  rocprofv3 --kernel-trace --stats -d /tmp/ocb$r -- python3 tools/diagnostics/bench_diag.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-flavour --no-larger-batch --no-native-share --hip-graph off > $O/bench_$r.json 2>/dev/null
  python3 tools/diagnostics/rocpd_stats.py /tmp/ocb$r/*/*_results.db $O/stats_$r.csv > /dev/null 2>&1
  echo "per CU: $r"; grep -E "out_conv_bwd|ar_update_loss_bwd" $O/stats_$r.csv | cut -c1-160
  python3 -c "import json; d=json.loads(open('$O/bench_$r.json').readlines()[-1]); print(d['value'], d['ms_per_step'])"
done
