#!/bin/bash
# Round 4: where the weight gradients' cost in the step comes from -- same box, one process per line, 20 timed steps each:
# usage: tools/diagnostics/r04_ab.sh <outfile> "ENV=VAL ENV=VAL" "..." ...   (an empty string = defaults)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
out=$1; shift; mkdir -p $(dirname $out)
: > $out
for cfg in "$@"; do
  line=$(env $cfg python3 bench.py --no-cpu-baseline --no-fp32-flavour --no-larger-batch --hip-graph off --steps 20 2>/dev/null | tail -1)
  python3 - "$cfg" "$line" >> $out <<'PY'
import json, sys
cfg, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line); r = d["roofline"]; s = d["step_ms"]
    print(f"{cfg or 'defaults':60s} ms/step {d['ms_per_step']:.3f}  min/median/max {s['min']:.3f}/{s['median']:.3f}/{s['max']:.3f}  conv {r['avg_launch_ms']*1e3:.1f}  "
          f"dgrad {(r.get('datagrad_avg_launch_ms_overlapped') or 0)*1e3:.1f}  wgrad {(r.get('wgrad_avg_launch_ms_overlapped') or 0)*1e3:.1f} us")
except Exception as e:
    print(f"{cfg:60s} FAILED {e} {line[-200:]}")
PY
done
cat $out
