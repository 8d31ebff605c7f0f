#!/bin/bash
# HBM traffic of a WHOLE optimizer step (round 4): FETCH_SIZE and WRITE_SIZE of every dispatch of a short bench run, separate --pmc passes
# with --kernel-trace only; summed per optimizer step (steps counted by the AdamW launches) by tools/diagnostics/pmc_step_sum.py.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04pmcstep; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-flavour --no-larger-batch --hip-graph off "$@" > $O/pmc_$c.log 2>&1
done
python3 tools/diagnostics/pmc_step_sum.py $O $O/pmc_traffic_step.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cat $O/pmc_traffic_step.json | head -60
