#!/bin/bash
# Round-6 CLOSING evidence run (one gpurun call; r06_profile.sh with the implicit-GEMM PMC passes one shape per process, as r06_final.sh): GPU suite + smoke, PMC passes first (separate --pmc runs with --kernel-trace only: the row conv
# kernel, the whole HalfUNet step, the implicit-GEMM convolution per shape + a known-size copy through its direct-to-LDS loads), then the bench
# lines that quote them (default line with other_configs, Titan shape, 500 steps), rocprofv3 kernel trace + stats of the bench command, the
# widened models' lines + kernel tables.  Every rocprofv3 line starts python3 directly and passes --no-native-share --no-other-configs (no second
# tracer, no child process under the profiler: ADVICE r5).  Outputs under gpurun_out/r06/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06; rm -rf $O; mkdir -p $O
python3 -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -12 > $O/gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
PM="FETCH_SIZE|WRITE_SIZE|SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE|SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
IFS='|'; for c in $PM; do
  n=$(echo $c | tr ' ' '_'); unset IFS
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 tools/diagnostics/conv_exp.py > $O/pmc_$n.log 2>&1
  IFS='|'
done; unset IFS
find $O -name "*.csv" -size +20M -delete
python3 tools/diagnostics/pmc_summary.py $O conv3x3_bf16_rows_kernel $O/pmc_traffic.json $O/pmc_rows > $O/pmc_summary.log 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_*
for shape in conv128 conv1024 linear; do
  IFS='|'; for c in $PM; do
    n=$(echo $c | tr ' ' '_'); unset IFS
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 tools/diagnostics/gemm_pmc.py $shape > $O/pmcg_${shape}_$n.log 2>&1
    IFS='|'
  done; unset IFS
  case $shape in
    conv128)  ALG='{"grid": 17072128}'; CM="conv3x3 128->128 on 2x128x128 (UNETR++ stage 0): input map 8.39 MB + output 8.39 MB + weight image 0.29 MB";;
    conv1024) ALG='{"grid": 27787264}'; CM="conv3x3 1024->1024 on 1x16x16, split-K (the kernel writes fp32 slabs, the epilogue is gemm_nt_reduce's): input 0.52 MB + weight image 18.9 MB + slabs 8.39 MB";;
    linear)   ALG='{"grid": 16809984}'; CM="Linear 32768x128 -> 128: every A row is read by exactly one tile through buffer_load ... lds (8.39 MB) + 8.39 MB out + 32 KB of weights: a known-size copy through the direct-to-LDS instruction -- ratio_fetch_as_counted ~ 0.76 and ratio_fetch_doubled ~ 1.02 mean the gfx950 FETCH_SIZE doubling DOES apply to it";;
  esac
  python3 tools/diagnostics/pmc_summary.py $O gemm_nt_kernel $O/pmc_traffic_gemm_nt_$shape.json $O/pmc_gemm_nt_$shape "$ALG" "gemm_pmc.py $shape: $CM" > $O/pmc_summary_gemm_nt_$shape.log 2>&1
  if [ $shape != linear ]; then
    python3 tools/diagnostics/pmc_summary.py $O gemm_tn_kernel $O/pmc_traffic_gemm_tn_$shape.json $O/pmc_gemm_tn_$shape '{"grid": 16777216}' "gemm_pmc.py $shape: weight gradient of the same convolution (x + dy: 16.8 MB at conv128, 1.05 MB at conv1024 -- the ratio of the latter is against the conv128 bytes and means nothing)" > $O/pmc_summary_gemm_tn_$shape.log 2>&1
  fi
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_*
done
mkdir -p gpurun_out/r06pmcstep
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r06pmcstep/pmc_$c -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-flavour --no-larger-batch --no-native-share --no-other-configs --hip-graph off > gpurun_out/r06pmcstep/pmc_$c.log 2>&1
done
python3 tools/diagnostics/pmc_step_sum.py gpurun_out/r06pmcstep $O/pmc_traffic_step.json > $O/pmc_step.log 2>&1
rm -rf gpurun_out/r06pmcstep
# the traffic files of THIS tree go where bench.py looks them up (profiles/, keyed on the kernel sources' hash) before the bench lines run
for f in pmc_traffic pmc_traffic_step pmc_traffic_gemm_nt_conv128 pmc_traffic_gemm_nt_conv1024 pmc_traffic_gemm_nt_linear pmc_traffic_gemm_tn_conv128 pmc_traffic_gemm_tn_conv1024; do [ -s $O/$f.json ] && cp $O/$f.json profiles/r06_$f.json; done
python3 bench.py > $O/halfunet_bf16_bench_default.json 2> $O/bench_default.err
python3 bench.py --grid 512 640 --features 21 --forcings 21 --border 10 --no-cpu-baseline --no-other-configs > $O/titan_shape_bench.json 2> $O/bench_titan.err
python3 bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-fp32-flavour --no-larger-batch --no-native-share --no-other-configs > $O/halfunet_bf16_bench_500_steps.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/raw -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-flavour --no-larger-batch --no-native-share --no-other-configs --hip-graph off > $O/halfunet_bf16_bench_under_rocprof.json 2> $O/trace.err
db=$(find $O/raw -name "*.db" | head -1)
python3 tools/diagnostics/rocpd_stats.py $db $O/halfunet_bf16_kernel_stats.csv $O/halfunet_bf16_one_step_trace.csv
python3 tools/diagnostics/step_timeline.py $db $O/timeline.csv > $O/halfunet_bf16_step_timeline.txt 2>&1
rm -rf $O/raw $O/timeline.csv
# the widened models
python3 bench.py --model SwinUNetR --cpu-seconds 5 > $O/swinunetr_bf16_bench.json 2>/dev/null
python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 5 --warmup 2 --cpu-seconds 5 > $O/unetrpp_bf16_bench.json 2>/dev/null
python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 5 --warmup 2 --no-cpu-baseline --unetrpp-block restated > $O/unetrpp_bf16_bench_restated_block.json 2>/dev/null
for m in GraphLam HiLAM HiLAMParallel; do python3 bench.py --model $m --no-cpu-baseline > $O/${m,,}_bf16_bench.json 2>/dev/null; done
python3 bench.py --model Identity --no-cpu-baseline > $O/identity_bench.json 2>/dev/null
rm -rf /tmp/ps /tmp/pu /tmp/ph /tmp/pg
rocprofv3 --kernel-trace --stats -d /tmp/ps -- python3 bench.py --model SwinUNetR --steps 5 --warmup 2 --no-cpu-baseline --no-native-share --no-other-configs --hip-graph off > /dev/null 2>&1
python3 tools/diagnostics/rocpd_stats.py /tmp/ps/*/*_results.db $O/swinunetr_bf16_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/pu -- python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 3 --warmup 1 --no-cpu-baseline --no-native-share --no-other-configs --hip-graph off > /dev/null 2>&1
python3 tools/diagnostics/rocpd_stats.py /tmp/pu/*/*_results.db $O/unetrpp_bf16_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/ph -- python3 bench.py --model HiLAM --steps 5 --warmup 2 --no-cpu-baseline --no-native-share --no-other-configs --hip-graph off > /dev/null 2>&1
python3 tools/diagnostics/rocpd_stats.py /tmp/ph/*/*_results.db $O/hilam_bf16_kernel_stats.csv
python3 tools/diagnostics/kernel_hist.py /tmp/ph/*/*_results.db row_mlp_bwd > $O/hilam_row_mlp_bwd_hist.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/pg -- python3 bench.py --model GraphLam --steps 5 --warmup 2 --no-cpu-baseline --no-native-share --no-other-configs --hip-graph off > /dev/null 2>&1
python3 tools/diagnostics/rocpd_stats.py /tmp/pg/*/*_results.db $O/graphlam_bf16_kernel_stats.csv
python3 tools/diagnostics/gemm_micro.py 2>&1 | grep -E "conv3x3|linear" > $O/gemm_micro.txt
cat $O/gpu_tests.txt $O/smoke.txt | tail -6
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06/*bench*.json")):
    try:
        d = json.loads(open(f).readlines()[-1])
        ns = (d.get("native_share") or {}).get("of_gpu_kernel_time")
        print(f.split("/")[-1], round(d["value"], 2), round(d["ms_per_step"], 3), d["roofline"].get("frac") if d.get("roofline") else None,
              (d["roofline"].get("step") or {}).get("frac") if d.get("roofline") else None, ns)
    except Exception as e:
        print(f, "FAILED", e)
PY
