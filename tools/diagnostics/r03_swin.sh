#!/bin/bash
# Round-3 SwinUNetR / UNetRPP bench lines + SwinUNetR kernel table after the rollout-input-format change (one gpurun call).
export TMPDIR=/tmp
O=gpurun_out/r03s; mkdir -p $O
python3 bench.py --model SwinUNetR --cpu-seconds 5 > $O/swinunetr_bf16_bench.json 2>/dev/null
python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 5 --warmup 2 --cpu-seconds 5 > $O/unetrpp_bf16_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d /tmp/ps -- python3 bench.py --model SwinUNetR --steps 5 --warmup 2 --no-cpu-baseline --hip-graph off > /dev/null 2>&1
python3 tools/diagnostics/rocpd_stats.py /tmp/ps/*/*_results.db $O/swinunetr_bf16_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/pu -- python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 3 --warmup 1 --no-cpu-baseline --hip-graph off > /dev/null 2>&1
python3 tools/diagnostics/rocpd_stats.py /tmp/pu/*/*_results.db $O/unetrpp_bf16_kernel_stats.csv
for f in $O/*.json; do python3 -c "
import json,sys; o=json.load(open('$f')); print('$f', round(o['value'],2), round(o['ms_per_step'],2), o['loss'], (o.get('roofline') or {}).get('frac'))"; done
