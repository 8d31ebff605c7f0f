#!/bin/bash
# round 6, call 8: first convolution as row launch + tail: tests, HalfUNet suite, step A/B through the diagnostic switch
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06h; mkdir -p $O
timeout 900 python3 -m pytest tests/test_first_conv_gpu.py -x -q > $O/test_first_conv.txt 2>&1; tail -8 $O/test_first_conv.txt
timeout 1800 python3 -m pytest tests/test_model_gpu.py tests/test_rollout_gpu.py tests/test_titan_shape_gpu.py tests/test_fused_tail_gpu.py tests/test_round2_gpu.py tests/test_unetrpp_gpu.py -x -q > $O/test_halfunet.txt 2>&1; tail -5 $O/test_halfunet.txt
Bn="--steps 20 --warmup 5 --no-cpu-baseline --no-fp32-flavour --no-larger-batch --no-native-share --no-other-configs"
for i in 1 2; do
  python3 tools/diagnostics/bench_diag.py $Bn > $O/split_$i.json 2>/dev/null
  P4C_FIRST_CONV_SPLIT=0 python3 tools/diagnostics/bench_diag.py $Bn > $O/one_launch_$i.json 2>/dev/null
done
python3 bench.py $Bn > $O/product.json 2>/dev/null
for f in $O/*.json; do echo $f $(python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"); done
