export SA_GUARD=0
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lv_forward_adjoint or robertson_forward or randomized or hermite or switched or error_test" 2>&1 | tail -2
run() { python bench.py --workload $1 --no-cpu-baseline --no-extra-configs --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 [$2]', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'fwd', round(d['roofline']['forward_kernel_ms'],3), 'bwd', round(d['roofline']['kernel_ms'],3), 'failed', d['config']['failed_instances'])"; }
run lv merged; run robertson merged
