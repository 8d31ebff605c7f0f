#!/bin/bash
# one-step kernel trace of the default bench workload: tools/diagnostics/trace_step.sh <tag> [ENV=VAL ...]
# -> gpurun_out/trace_<tag>/{kernel_stats.csv, one_step_trace.csv, bench.json}
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/trace_$tag; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/raw -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-flavour --no-larger-batch --hip-graph off > $O/bench.json 2> $O/trace.err
db=$(find $O/raw -name "*.db" | head -1)
python3 tools/diagnostics/rocpd_stats.py $db $O/kernel_stats.csv $O/one_step_trace.csv
python3 tools/diagnostics/step_timeline.py $db $O/timeline.csv > $O/timeline.txt 2>&1
rm -rf $O/raw
