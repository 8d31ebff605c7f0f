for m in "SwinUNetR" "UNetRPP --strategy diff_ar --pred-steps 6" "GraphLam" "HiLAM"; do
python3 bench.py --model $m --steps 3 --warmup 2 --no-cpu-baseline 2>/tmp/err.txt | python3 -c "
import json,sys; o=json.loads(sys.stdin.read()); print('$m', o['value'], o['ms_per_step'], o['loss'], o['config']['hip_graph'], o['config']['hip_graph_check'])"; grep "bench:" /tmp/err.txt | head -3
done
