#!/bin/bash
# Final round-2 evidence on one box: full GPU suite, smoke, default + accumulate-10 bench lines, widened-model bench lines + kernel
# tables, the Linear micro-benchmark.  Outputs under gpurun_out/r02f/ (copied into profiles/ by hand).
export TMPDIR=/tmp
O=gpurun_out/r02f; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -2 > $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $O/gpu_tests.txt
python3 bench.py > $O/halfunet_bf16_bench_default.json 2>/dev/null
python3 bench.py --accumulate 10 > $O/halfunet_bf16_bench_accumulate10.json 2>/dev/null
python3 bench.py --model SwinUNetR --cpu-seconds 5 > $O/swinunetr_bf16_bench.json 2>/dev/null
python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 5 --warmup 2 --cpu-seconds 5 > $O/unetrpp_bf16_bench.json 2>/dev/null
for m in GraphLam HiLAM HiLAMParallel; do python3 bench.py --model $m --no-cpu-baseline > $O/$(echo $m | tr 'A-Z' 'a-z')_bf16_bench.json 2>/dev/null; done
python3 tools/diagnostics/linear_micro.py 2>&1 | grep -v amdgpu.ids > $O/linear_micro.txt
rocprofv3 --kernel-trace --stats -d /tmp/ps -- python3 bench.py --model SwinUNetR --steps 5 --warmup 2 --no-cpu-baseline --hip-graph off > /dev/null 2>&1
python3 tools/diagnostics/rocpd_stats.py /tmp/ps/*/*_results.db $O/swinunetr_bf16_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/pu -- python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 3 --warmup 1 --no-cpu-baseline --hip-graph off > /dev/null 2>&1
python3 tools/diagnostics/rocpd_stats.py /tmp/pu/*/*_results.db $O/unetrpp_bf16_kernel_stats.csv
cat $O/gpu_tests.txt
for f in $O/*bench*.json; do python3 -c "
import json,sys; o=json.load(open('$f')); print('$f', round(o['value'],2), round(o['ms_per_step'],3), o['loss'], round((o.get('roofline') or {}).get('frac',0),3), (o.get('larger_batch') or {}).get('value'))"; done
