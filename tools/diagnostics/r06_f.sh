#!/bin/bash
# round 6, call 6: where the UNETR++ / SwinUNETR steps go now: ATen device time by autograd node, kernel tables (eager, 3 steps)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06f; mkdir -p $O
timeout 600 python3 -m pytest tests/test_bench_contract_gpu.py -x -q > $O/test_contract.txt 2>&1; tail -3 $O/test_contract.txt
python3 tools/diagnostics/op_stacks.py UNetRPP > $O/op_stacks_unetrpp.txt 2>&1; head -45 $O/op_stacks_unetrpp.txt
python3 tools/diagnostics/op_stacks.py SwinUNetR > $O/op_stacks_swin.txt 2>&1; head -40 $O/op_stacks_swin.txt
bash tools/diagnostics/model_stats.sh r06f_unetrpp --model UNetRPP --strategy diff_ar --pred-steps 6 --no-native-share --unetrpp-block restated > $O/unetrpp_table.txt 2>&1; cat $O/unetrpp_table.txt
bash tools/diagnostics/model_stats.sh r06f_swin --model SwinUNetR --no-native-share > $O/swin_table.txt 2>&1; cat $O/swin_table.txt
cp gpurun_out/stats_r06f_*.csv $O/
