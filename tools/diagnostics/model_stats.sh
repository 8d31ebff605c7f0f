#!/bin/bash
# kernel table of one model's eager training steps: tools/diagnostics/model_stats.sh <tag> <bench.py args ...>
# -> gpurun_out/stats_<tag>.csv + native share / top kernels on stdout
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
tag=$1; shift
rm -rf /tmp/ps_$tag
rocprofv3 --kernel-trace --stats -d /tmp/ps_$tag -- python3 bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --hip-graph off > gpurun_out/stats_$tag.json 2> /dev/null
python3 tools/diagnostics/rocpd_stats.py $(find /tmp/ps_$tag -name "*.db" | head -1) gpurun_out/stats_$tag.csv > /dev/null
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/stats_$tag.csv')))
tot=sum(float(r['TotalDurationUs']) for r in rows); nat=sum(float(r['TotalDurationUs']) for r in rows if 'p4c' in r['Name'])
print(f"$tag: kernel time {tot/1e3:.1f} ms in the profiled window, native {100*nat/tot:.1f} %")
for r in rows[:28]:
    print(f"  {float(r['TotalDurationUs'])/tot*100:5.1f}%  {r['Calls']:>6} x {float(r['AverageUs']):8.1f} us  {r['Name'][:150]}")
PY
