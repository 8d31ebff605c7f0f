#!/bin/bash
# Round-5 evidence for UNETR++ (BASELINE configuration 5): bench line (6-step diff_ar) + kernel table of an eager run.
export TMPDIR=/tmp
O=gpurun_out/r05u; mkdir -p $O
python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 5 --warmup 2 --cpu-seconds 5 > $O/unetrpp_bf16_bench.json 2>$O/bench.err
rocprofv3 --kernel-trace --stats -d /tmp/pu -- python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 3 --warmup 1 --no-cpu-baseline --hip-graph off > /dev/null 2>&1
python3 tools/diagnostics/rocpd_stats.py /tmp/pu/*/*_results.db $O/unetrpp_bf16_kernel_stats.csv
python3 -c "
import json; o=json.load(open('$O/unetrpp_bf16_bench.json')); print(round(o['value'],2), round(o['ms_per_step'],2), o['loss'], o['config'].get('hip_graph'), o['config'].get('host_loop_ms_per_step'))"
