#!/bin/bash
# round 6: the whole GPU suite + smoke on the current tree
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r06t
python3 -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -15 > gpurun_out/r06t/gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke > gpurun_out/r06t/smoke.txt
cat gpurun_out/r06t/gpu_tests.txt gpurun_out/r06t/smoke.txt
