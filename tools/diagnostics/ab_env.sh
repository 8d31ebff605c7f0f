#!/bin/bash
# same-box A/B of an environment switch: scratch/ab_env.sh VAR valA valB
for rep in 1 2; do
for v in "$2" "$3"; do
  env $1=$v python bench.py --no-cpu-baseline --steps 15 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
k = d['kernel_ms']
print('$1=$v', 'ms/step %.3f' % d['ms_per_step'], 'fwd %.4f bwd %.4f' % (k['p4c_halfunet_forward']['avg_ms'], k['p4c_halfunet_backward']['avg_ms']), 'loss %.4f' % d['loss'])
"
done
done
