export SA_GUARD=0
run() { python bench.py --workload $1 --no-cpu-baseline --no-extra-configs --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 [$2]', round(d['value']), 'fwd', round(d['roofline']['forward_kernel_ms'],3), 'bwd', round(d['roofline']['kernel_ms'],3), 'failed', d['config']['failed_instances'])"; }
D="-mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -disable-machine-licm -mllvm -split-spill-mode=size"
for w in lv robertson; do
run $w base
SA_CLANG_FLAGS="$D -mllvm -greedy-regclass-priority-trumps-globalness=1" run $w regclassprio
SA_CLANG_FLAGS="$D -mllvm -amdgpu-use-amdgpu-trackers=1" run $w trackers
SA_CLANG_FLAGS="$D -mllvm -amdgpu-schedule-relaxed-occupancy=1" run $w relaxedocc
SA_CLANG_FLAGS="$D -mllvm -amdgpu-vgpr-index-mode=0 -mllvm -enable-post-misched=0" run $w nopostsched
SA_CLANG_FLAGS="-mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -disable-machine-licm" run $w nosplitsize
done
