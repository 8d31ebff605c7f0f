#!/bin/bash
# up_bwd_x4 on the matrix cores against the vector-ALU kernel (P4C_UPBWD_VALU=1, diagnostic library): parity tests, per-kernel durations of
# the HalfUNet bench under rocprofv3, the bench line on the product library.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05ub; mkdir -p $O
python3 -m pytest tests/test_upbwd_gpu.py -m gpu -x -q 2>&1 | tail -8
for r in valu mfma; do
  if [ $r = valu ]; then export P4C_UPBWD_VALU=1; else unset P4C_UPBWD_VALU; fi
  rocprofv3 --kernel-trace --stats -d /tmp/ub$r -- python3 tools/diagnostics/bench_diag.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-flavour --no-larger-batch --no-native-share --hip-graph off > $O/bench_$r.json 2>/dev/null
  python3 tools/diagnostics/rocpd_stats.py /tmp/ub$r/*/*_results.db $O/stats_$r.csv > /dev/null 2>&1
  echo "$r"; grep -E "up_bwd_x4|enc_out_bwd_blk" $O/stats_$r.csv | cut -c1-200
done
unset P4C_UPBWD_VALU
for i in 1 2; do python3 bench.py --no-cpu-baseline --no-fp32-flavour --no-larger-batch --no-native-share | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done
