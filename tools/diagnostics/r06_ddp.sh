#!/bin/bash
# round 6: the in-place gradient exchange tests + the external-event probe
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r06k
timeout 120 python3 tools/diagnostics/r06_external_event.py > gpurun_out/r06k/external_event.txt 2>&1
timeout 1500 python3 -m pytest tests/test_dist_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/r06k/dist_tests.txt
cat gpurun_out/r06k/external_event.txt
tail -15 gpurun_out/r06k/dist_tests.txt
