#!/bin/bash
# Round-4 evidence for the widened models (one gpurun call): bench lines + kernel tables of SwinUNetR and UNetRPP, GNN bench lines.
export TMPDIR=/tmp
O=gpurun_out/r04m; mkdir -p $O
python3 bench.py --model SwinUNetR --cpu-seconds 5 > $O/swinunetr_bf16_bench.json 2>/dev/null
python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 5 --warmup 2 --cpu-seconds 5 > $O/unetrpp_bf16_bench.json 2>/dev/null
for m in GraphLam HiLAM HiLAMParallel; do python3 bench.py --model $m --no-cpu-baseline > $O/${m}_bf16_bench.json 2>/dev/null; done
rocprofv3 --kernel-trace --stats -d /tmp/ph -- python3 bench.py --model HiLAM --steps 5 --warmup 2 --no-cpu-baseline --hip-graph off > /dev/null 2>&1
python3 tools/diagnostics/rocpd_stats.py /tmp/ph/*/*_results.db $O/hilam_bf16_kernel_stats.csv
python3 bench.py --model Identity --no-cpu-baseline > $O/identity_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d /tmp/ps -- python3 bench.py --model SwinUNetR --steps 5 --warmup 2 --no-cpu-baseline --hip-graph off > /dev/null 2>&1
python3 tools/diagnostics/rocpd_stats.py /tmp/ps/*/*_results.db $O/swinunetr_bf16_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/pu -- python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 3 --warmup 1 --no-cpu-baseline --hip-graph off > /dev/null 2>&1
python3 tools/diagnostics/rocpd_stats.py /tmp/pu/*/*_results.db $O/unetrpp_bf16_kernel_stats.csv
for f in $O/*.json; do python3 -c "
import json,sys; o=json.load(open('$f')); print('$f', round(o['value'],2), round(o['ms_per_step'],2), o['loss'], (o.get('roofline') or {}).get('frac'))"; done
