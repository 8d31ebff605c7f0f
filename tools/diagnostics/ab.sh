#!/bin/bash
# same-box A/B: runs bench.py alternately with the libraries given as arguments
for rep in 1 2; do
for lib in "$@"; do
  P4C_LIB_PATH=$GRAFT_REPO_ROOT/tools/diagnostics/libs/$lib python bench.py --dtype bf16 --no-cpu-baseline --steps 15 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
k = d['kernel_ms']
print('$lib', 'ms/step %.3f' % d['ms_per_step'], 'fwd %.4f bwd %.4f bx %.4f' % (k['p4c_halfunet_forward']['avg_ms'], k['p4c_halfunet_backward']['avg_ms'], k['p4c_build_x']['avg_ms']), 'conv %.4f wgrad %.4f' % (d['roofline']['avg_launch_ms'], d['roofline']['wgrad_avg_launch_ms']))
"
done
done
