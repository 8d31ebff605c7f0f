#!/bin/bash
# Round-2 evidence run (one gpurun call): default bench line, rocprofv3 kernel trace of the same command, PMC passes of the
# roofline kernel (separate --pmc runs with --kernel-trace only, as gpurun requires).  Outputs under gpurun_out/r02/.
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --accumulate 10 --no-cpu-baseline --no-fp32-flavour > $O/bench_accumulate10.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-flavour --hip-graph off > $O/bench_under_rocprof.json 2> $O/trace.err
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 tools/diagnostics/conv_exp.py > $O/pmc_$n.log 2>&1
done
find $O -name "*.csv" -size +20M -delete
ls -R $O | head -60
