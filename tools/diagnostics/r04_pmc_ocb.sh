#!/bin/bash
# PMC passes of the fused output-convolution backward (separate --pmc runs with --kernel-trace only) -> gpurun_out/r04pmc/
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04pmc; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 tools/diagnostics/ocb_pmc.py > $O/pmc_$n.log 2>&1
done
find $O -name "*.csv" -size +20M -delete
python3 tools/diagnostics/pmc_summary.py $O out_conv_bwd_kernel $O/pmc_traffic_out_conv_bwd.json $O/pmc_out_conv_bwd 201326592 > $O/pmc_summary.log 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_*
cat $O/pmc_traffic_out_conv_bwd.json
