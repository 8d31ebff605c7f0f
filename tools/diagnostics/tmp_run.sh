for i in 1 2; do
python3 bench.py 2>/dev/null | python3 -c "
import json,sys; o=json.loads(sys.stdin.read()); print('tuned-gemm env on ', round(o['value'],2), round(o['ms_per_step'],3), o['step_ms']['min'], o['roofline']['frac'])"
P4C_NO_TUNED_GEMMS=1 python3 bench.py 2>/dev/null | python3 -c "
import json,sys; o=json.loads(sys.stdin.read()); print('tuned-gemm env off', round(o['value'],2), round(o['ms_per_step'],3), o['step_ms']['min'], o['roofline']['frac'])"
done
