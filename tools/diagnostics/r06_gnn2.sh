#!/bin/bash
# round 6, call 2: new parity tests (PatchMerging golden, side-stream switch) + HiLAM kernel table and per-grid histogram on the grouped routes
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06b; mkdir -p $O
timeout 900 python3 -m pytest tests/test_swin_golden_gpu.py tests/test_model_gpu.py -x -q -k "merging or side_stream" > $O/tests.txt 2>&1
tail -4 $O/tests.txt
bash tools/diagnostics/model_stats.sh r06b_hilam --model HiLAM --no-native-share > $O/hilam_table.txt 2>&1
cp gpurun_out/stats_r06b_hilam.csv $O/ 2>/dev/null
cat $O/hilam_table.txt
db=$(find /tmp/ps_r06b_hilam -name "*.db" | head -1)
for k in row_mlp_bwd row_mlp_fwd node_proj_fwd node_proj_dgrad node_proj_wgrad grad_reduce_batch segment_sum edge_gather; do
  python3 tools/diagnostics/kernel_hist.py $db $k | head -12
done > $O/hilam_hist.txt 2>&1
cat $O/hilam_hist.txt
