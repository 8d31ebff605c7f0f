#!/bin/bash
# Round-3 UNetRPP evidence after the matrix-core EPA kernels (one gpurun call): bench line, kernel table, per-launch times of the
# tall-skinny kernels on the matrix cores and on the VALU kernels they replace.
export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 5 --warmup 2 --cpu-seconds 5 > $O/unetrpp_bf16_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d /tmp/pu -- python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 3 --warmup 1 --no-cpu-baseline --hip-graph off > /dev/null 2>&1
python3 tools/diagnostics/rocpd_stats.py /tmp/pu/*/*_results.db $O/unetrpp_bf16_kernel_stats.csv
{ for hd in 16 4; do echo "== $hd heads, matrix-core kernels (default)"; HEADS=$hd python3 tools/diagnostics/ts_micro.py
  echo "== $hd heads, VALU kernels (P4C_TS_NO_MFMA=1; wider than 64 columns: in chunks)"; HEADS=$hd P4C_TS_NO_MFMA=1 python3 tools/diagnostics/ts_micro.py; done; } > $O/ts_micro.txt 2>/dev/null
python3 -c "
import json; o=json.load(open('$O/unetrpp_bf16_bench.json')); print(round(o['value'],3), round(o['ms_per_step'],2), o['loss'], o['config']['hip_graph_check'])"
