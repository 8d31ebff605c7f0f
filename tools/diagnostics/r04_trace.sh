#!/bin/bash
# rocprofv3 kernel trace of a short bench run: usage tools/diagnostics/r04_trace.sh <outdir> [bench.py arguments]
# -> <outdir>/kernel_stats.csv, one_step_trace.csv, timeline.txt
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$1; shift; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/raw -- python3 bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-flavour --no-larger-batch --hip-graph off > $O/bench.json 2> $O/trace.err
db=$(find $O/raw -name "*.db" | head -1)
python3 tools/diagnostics/rocpd_stats.py $db $O/kernel_stats.csv $O/one_step_trace.csv
python3 tools/diagnostics/step_timeline.py $db $O/timeline.csv > $O/timeline.txt 2>&1
rm -rf $O/raw $O/timeline.csv
head -40 $O/kernel_stats.csv
