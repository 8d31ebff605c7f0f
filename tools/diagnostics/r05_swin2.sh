#!/bin/bash
# SwinUNETR: the stage padded once (swinunetr.padded_stage) against MONAI's per-block pad / crop (P4C_SWIN_PAD_PER_BLOCK=1): bench line x 3 each
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
for v in 1 0; do
  if [ $v = 1 ]; then export P4C_SWIN_PAD_PER_BLOCK=1; else unset P4C_SWIN_PAD_PER_BLOCK; fi
  python3 bench.py --model SwinUNetR --no-cpu-baseline --no-native-share --steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('per-block' if $v else 'per-stage', d['ms_per_step'], d['config'].get('hip_graph'))"
done
done
