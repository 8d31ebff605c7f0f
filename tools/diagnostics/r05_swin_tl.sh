#!/bin/bash
# SwinUNETR step replayed from its HIP graph under rocprofv3 --kernel-trace: busy time per queue and the gaps of one step
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05sw; mkdir -p $O
rocprofv3 --kernel-trace -d /tmp/swg -- python3 bench.py --model SwinUNetR --steps 5 --warmup 2 --no-cpu-baseline --no-native-share --hip-graph on > $O/bench_graph.json 2>/dev/null
db=$(find /tmp/swg -name "*.db" | head -1)
python3 tools/diagnostics/step_timeline.py $db $O/tl.csv weighted_loss_final 2>&1 | head -60
python3 -c "import json; d=json.loads(open('$O/bench_graph.json').readlines()[-1]); print(d['ms_per_step'])"
