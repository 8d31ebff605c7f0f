#!/bin/bash
# round 6, call 4: the published UNETR++ block (tests, A/B against the restated block at the bench size) + the default bench line with other_configs
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06d; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_unetrpp_gpu.py -x -q > $O/test_unetrpp.txt 2>&1
tail -6 $O/test_unetrpp.txt
timeout 1500 python3 -m pytest tests/test_bench_contract_gpu.py tests/test_widen_gpu.py -x -q -k "contract or other_baseline or stand_ins or widened_models" > $O/test_contract.txt 2>&1
tail -6 $O/test_contract.txt
U="--model UNetRPP --strategy diff_ar --pred-steps 6 --steps 5 --warmup 2 --no-cpu-baseline"
python3 bench.py $U --unetrpp-block restated > $O/unetrpp_restated.json 2> $O/unetrpp_restated.err
python3 bench.py $U --unetrpp-block published-nodrop > $O/unetrpp_published_nodrop.json 2> $O/unetrpp_published_nodrop.err
python3 bench.py $U > $O/unetrpp_published.json 2> $O/unetrpp_published.err
python3 bench.py > $O/halfunet_default.json 2> $O/halfunet_default.err
for f in $O/*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config'].get('hip_graph'), d['config'].get('hip_graph_check'), json.dumps(d.get('other_configs'))[:1500])
except Exception as e: print('ERR', e)
"; done
tail -3 $O/*.err
