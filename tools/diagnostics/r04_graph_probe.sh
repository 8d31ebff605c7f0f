#!/bin/bash
# Round 4: why does the HIP-graph replay of the HalfUNet step take longer than eager launching?  One kernel trace of each launch mode
# (same box, same process arguments), exported as per-dispatch timelines; tools/diagnostics/graph_vs_eager.py compares kernel
# durations and the gaps between consecutive kernels of the main queue.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r04g; mkdir -p $O
for mode in off on; do
  rocprofv3 --kernel-trace --stats -d $O/raw_$mode -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-flavour --no-larger-batch --hip-graph $mode > $O/bench_$mode.json 2> $O/trace_$mode.err
  db=$(find $O/raw_$mode -name "*.db" | head -1)
  python3 tools/diagnostics/step_timeline.py $db $O/timeline_$mode.csv > $O/timeline_$mode.txt 2>&1
  rm -rf $O/raw_$mode
done
python3 tools/diagnostics/graph_vs_eager.py $O/timeline_off.csv $O/timeline_on.csv > $O/graph_vs_eager.txt 2>&1
cat $O/graph_vs_eager.txt
