#!/bin/bash
# same-box A/B (round 2): runs bench.py alternately with the libraries given as arguments (names under tools/diagnostics/libs/)
for rep in 1 2 3; do
for lib in "$@"; do
  P4C_LIB_PATH=$GRAFT_REPO_ROOT/tools/diagnostics/libs/$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-flavour --hip-graph off 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$lib', 'ms/step %.3f' % d['ms_per_step'], 'median %.3f' % d['step_ms']['median'], 'conv %.4f' % d['roofline']['avg_launch_ms'], 'loss', d['loss'])
"
done
done
