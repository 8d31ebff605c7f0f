export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04g; mkdir -p $O
export P4C_SIDE_STREAM=0
for mode in off on; do
  rocprofv3 --kernel-trace --stats -d $O/raw1_$mode -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-flavour --no-larger-batch --hip-graph $mode > $O/bench1_$mode.json 2> $O/trace1_$mode.err
  db=$(find $O/raw1_$mode -name "*.db" | head -1)
  python3 tools/diagnostics/step_timeline.py $db $O/timeline1_$mode.csv > $O/timeline1_$mode.txt 2>&1
  rm -rf $O/raw1_$mode
done
python3 tools/diagnostics/graph_vs_eager.py $O/timeline1_off.csv $O/timeline1_on.csv > $O/graph_vs_eager_one_stream.txt 2>&1
head -8 $O/graph_vs_eager_one_stream.txt
for mode in off on; do python3 bench.py --steps 20 --no-cpu-baseline --no-fp32-flavour --no-larger-batch --hip-graph $mode 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('one stream, hip-graph $mode:', round(d['ms_per_step'],3), d['step_ms']['median'])"; done
unset P4C_SIDE_STREAM
for mode in off on; do python3 bench.py --steps 20 --no-cpu-baseline --no-fp32-flavour --no-larger-batch --hip-graph $mode 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('two streams, hip-graph $mode:', round(d['ms_per_step'],3), d['step_ms']['median'])"; done
