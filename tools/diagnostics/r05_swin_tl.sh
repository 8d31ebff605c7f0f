#!/bin/bash
# SwinUNETR step replayed from its HIP graph under rocprofv3 --kernel-trace: span / busy time / gaps of one step, stage padded once against
# padded in every block
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05sw; mkdir -p $O
for v in 1 0; do
  if [ $v = 1 ]; then export P4C_SWIN_PAD_PER_BLOCK=1; else unset P4C_SWIN_PAD_PER_BLOCK; fi
  rm -rf /tmp/swg
  rocprofv3 --kernel-trace -d /tmp/swg -- python3 bench.py --model SwinUNetR --steps 5 --warmup 2 --no-cpu-baseline --no-native-share --hip-graph on > $O/bench_graph_$v.json 2>/dev/null
  db=$(find /tmp/swg -name "*.db" | head -1)
  echo "pad per block: $v"
  python3 tools/diagnostics/step_timeline.py $db $O/tl_$v.csv build_x_flat 2>&1 | sed -n 2,12p
  python3 -c "import json; d=json.loads(open('$O/bench_graph_$v.json').readlines()[-1]); print(d['ms_per_step'])"
done
