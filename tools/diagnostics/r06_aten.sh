#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r06l
timeout 900 python3 tools/diagnostics/r06_aten_sources.py UNetRPP 6 diff_ar 60 > gpurun_out/r06l/aten_unetrpp.txt 2>&1
timeout 600 python3 tools/diagnostics/r06_aten_sources.py SwinUNetR 3 scaled_ar 40 > gpurun_out/r06l/aten_swin.txt 2>&1
tail -5 gpurun_out/r06l/aten_unetrpp.txt
