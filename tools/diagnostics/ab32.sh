#!/bin/bash
for rep in 1 2; do
for lib in "$@"; do
  P4C_LIB_PATH=$GRAFT_REPO_ROOT/tools/diagnostics/libs/$lib python bench.py --dtype f32 --no-cpu-baseline --steps 8 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$lib', 'ms/step %.3f' % d['ms_per_step'], 'conv frac %.4f  %.4f ms' % (d['roofline']['frac'], d['roofline']['avg_launch_ms']), 'wgrad %.4f' % d['roofline']['wgrad_kernel']['avg_launch_ms'])
"
done
done
