#!/bin/bash
# Round-4 evidence run (one gpurun call): default bench line, Titan-shape line, accumulate-10 line, 500-step robustness run, rocprofv3
# kernel trace of the bench command, PMC passes of the row conv kernel and of the row-streaming weight-gradient kernel (separate
# --pmc runs with --kernel-trace only, as gpurun requires).  Outputs under gpurun_out/r04/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r04; mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --grid 512 640 --features 21 --forcings 21 --border 10 > $O/bench_titan.json 2> $O/bench_titan.err
python3 bench.py --accumulate 10 --no-cpu-baseline --no-fp32-flavour --no-larger-batch > $O/bench_accumulate10.json 2>/dev/null
python3 bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-fp32-flavour --no-larger-batch > $O/bench_500_steps.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/raw -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-flavour --no-larger-batch --hip-graph off > $O/bench_under_rocprof.json 2> $O/trace.err
db=$(find $O/raw -name "*.db" | head -1)
python3 tools/diagnostics/rocpd_stats.py $db $O/kernel_stats.csv $O/one_step_trace.csv
python3 tools/diagnostics/step_timeline.py $db $O/timeline.csv > $O/timeline.txt 2>&1
rm -rf $O/raw
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 tools/diagnostics/conv_exp.py > $O/pmc_$n.log 2>&1
done
find $O -name "*.csv" -size +20M -delete
python3 tools/diagnostics/pmc_summary.py $O conv3x3_bf16_rows_kernel $O/pmc_traffic.json $O/pmc_rows > $O/pmc_summary.log 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_*
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 tools/diagnostics/wgrad_pmc.py > $O/pmcw_$n.log 2>&1
done
find $O -name "*.csv" -size +20M -delete
python3 tools/diagnostics/pmc_summary.py $O conv3x3_wgrad_bf16_rows_kernel $O/pmc_traffic_wgrad.json $O/pmc_wgrad_rows > $O/pmc_summary_wgrad.log 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_*
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
ls -R $O | head -60
