#!/bin/bash
# Round-5 evidence for SwinUNETR (BASELINE configuration 3): bench line + kernel table of an eager run.
export TMPDIR=/tmp
O=gpurun_out/r05s; mkdir -p $O
python3 bench.py --model SwinUNetR --cpu-seconds 5 > $O/swinunetr_bf16_bench.json 2>$O/bench.err
rocprofv3 --kernel-trace --stats -d /tmp/ps -- python3 bench.py --model SwinUNetR --steps 5 --warmup 2 --no-cpu-baseline --hip-graph off > /dev/null 2>&1
python3 tools/diagnostics/rocpd_stats.py /tmp/ps/*/*_results.db $O/swinunetr_bf16_kernel_stats.csv
python3 -c "
import json; o=json.load(open('$O/swinunetr_bf16_bench.json')); print(round(o['value'],2), round(o['ms_per_step'],2), o['loss'], o['config'].get('hip_graph'), o['config'].get('host_loop_ms_per_step'))"
