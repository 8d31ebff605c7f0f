#!/bin/bash
# round 6, call 9: in-kernel instance-norm finalize (tests + Swin / UNETR++ step A/B); the first-conv plan test that failed in the full suite
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06i; mkdir -p $O
timeout 900 python3 -m pytest tests/test_first_conv_gpu.py -x -q > $O/test_first_conv.txt 2>&1; grep -E "^E  |passed|failed" $O/test_first_conv.txt | head -20
timeout 900 python3 -m pytest tests/test_inorm_fin_gpu.py -x -q > $O/test_inorm.txt 2>&1; grep -E "^E  |passed|failed" $O/test_inorm.txt | head -20
timeout 1800 python3 -m pytest tests/test_widen_gpu.py tests/test_unetrpp_gpu.py tests/test_gemm_gpu.py tests/test_swin_golden_gpu.py -x -q -k "swin or unetrpp or instance or batch_norm or group_norm or graphed" > $O/test_models.txt 2>&1; tail -3 $O/test_models.txt
cat > /tmp/nofuse.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
from py4cast_amd import ops_inorm as ON
ON.FUSED_FINALIZE = False
bench.main()
PY
S="--model SwinUNetR --steps 8 --warmup 3 --no-cpu-baseline --no-native-share"
U="--model UNetRPP --strategy diff_ar --pred-steps 6 --steps 5 --warmup 2 --no-cpu-baseline --no-native-share --unetrpp-block restated"
python3 bench.py $S > $O/swin_fused.json 2>/dev/null
python3 /tmp/nofuse.py $S > $O/swin_two_launch.json 2>/dev/null
python3 bench.py $S > $O/swin_fused2.json 2>/dev/null
python3 bench.py $U > $O/unetrpp_fused.json 2>/dev/null
python3 /tmp/nofuse.py $U > $O/unetrpp_two_launch.json 2>/dev/null
for f in $O/*.json; do echo $f $(python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'])"); done
