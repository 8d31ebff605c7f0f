#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06r}
mkdir -p $O gpurun_out/r06l
timeout 1500 python3 -m pytest tests/test_unetrpp_gpu.py tests/test_widen_gpu.py tests/test_gemm_gpu.py tests/test_bench_size_gpu.py -q 2>&1 | tail -25 > $O/tests.txt
tail -6 $O/tests.txt
timeout 900 python3 tools/diagnostics/r06_aten_sources.py UNetRPP 6 diff_ar 60 > $O/aten_unetrpp.txt 2>&1
timeout 600 python3 tools/diagnostics/r06_aten_sources.py SwinUNetR 3 scaled_ar 45 > $O/aten_swin.txt 2>&1
