export SA_GUARD=0
(time python -m pytest tests/test_gpu_parity.py tests/test_multi_device.py tests/test_gpu_fullsize.py tests/test_forward_sens.py -m gpu -q -x --timeout 1500 -k "network or eight or handles or sens or seir" 2>&1 | tail -6) 2>&1 | tail -12
python bench.py --no-cpu-baseline --steps 10 > gpurun_out/bench_d.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/bench_d.json').read().strip().splitlines()[-1]); print('lv', round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms'], d.get('host_api',{}).get('ms_per_step')); print({k:(round(v.get('solves_per_s',0)), round(v.get('forward_kernel_ms',0),2), round(v.get('backward_kernel_ms',0),2)) for k,v in d.get('configs',{}).items()})"
for a in "lv 65536" "robertson 65536" "seir 16384"; do python tools/bench_sens.py $a 2>&1 | tail -1; done
