#!/bin/bash
# PyTorch TunableOp pass over the GEMM shapes of the bench workloads (library GEMMs only: Linear layers, token-axis projections,
# pixel-block convolutions).  Writes one result file per workload under gpurun_out/tunable/; tools/diagnostics/merge_tunable.py
# merges them into py4cast_amd/tuning/tunableop_gfx950.csv, which the package loads with tuning switched off.
mkdir -p gpurun_out/tunable
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=100 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=10
export P4C_NO_TUNED_GEMMS=1
run() { name=$1; shift; PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunable/${name}_%d.csv python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --hip-graph off 2>/dev/null | python3 -c "
import json,sys; o=json.loads(sys.stdin.read()); print('$name', round(o['ms_per_step'],2), o['loss'])"; }
run swinunetr --model SwinUNetR
run unetrpp --model UNetRPP --strategy diff_ar --pred-steps 6
run graphlam --model GraphLam
run hilam --model HiLAM
run hilamparallel --model HiLAMParallel
run halfunet
wc -l gpurun_out/tunable/*.csv
