#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06v; mkdir -p $O
timeout 600 python3 tools/diagnostics/r06_aten_sources.py GraphLam 3 scaled_ar 40 > $O/aten_graphlam.txt 2>&1
timeout 600 python3 tools/diagnostics/r06_aten_sources.py HiLAM 3 scaled_ar 40 > $O/aten_hilam.txt 2>&1
grep -E "device kernel time" $O/*.txt
