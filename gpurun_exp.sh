export SA_GUARD=0
run() { python bench.py --workload seir --no-cpu-baseline --no-extra-configs --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('seir [$1]', round(d['value']), 'fwd', round(d['roofline']['forward_kernel_ms'],3), 'bwd', round(d['roofline']['kernel_ms'],3), 'failed', d['config']['failed_instances'])"; }
W="-mllvm -disable-machine-licm -mllvm -machine-sink-split=0 -mllvm -split-spill-mode=size -mllvm -misched-cluster=0"
run default
SA_CLANG_FLAGS="$W -mllvm -amdgpu-use-amdgpu-trackers=1" run trackers
SA_CLANG_FLAGS="-mllvm -disable-machine-licm -mllvm -machine-sink-split=0 -mllvm -misched-cluster=0" run nosplitsize
SA_CLANG_FLAGS="$W -mllvm -greedy-regclass-priority-trumps-globalness=1" run regclassprio
SA_CLANG_FLAGS="-mllvm -disable-machine-licm -mllvm -machine-sink-split=0 -mllvm -misched-cluster=0 -mllvm -amdgpu-use-amdgpu-trackers=1" run nosplit_trackers
