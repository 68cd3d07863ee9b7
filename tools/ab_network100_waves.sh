#!/bin/bash
# config 5: wavefronts per workgroup (= per instance) of the workgroup-per-instance mapping: 4 (default), 2, 8.
#   bash tools/ab_network100_waves.sh > gpurun_out/r06_network100_waves.txt        (MI355X box; code objects pre-built)
echo "# python bench.py --workload network100 (B = 1 024, fwd + adjoint), SA_KERNEL_DEFINES=-DSA_WAVES=<w>: solves/s, ms per step, kernel ms (r06, MI355X)"
for w in default 2 8; do
    if [ "$w" = default ]; then defs=""; else defs="-DSA_WAVES=$w"; fi
    SA_GUARD=0 SA_KERNEL_DEFINES="$defs" timeout 900 python bench.py --workload network100 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 | \
      python -c "import sys, json; d = json.loads(sys.stdin.read()); r = d['roofline']; print('SA_WAVES=$w: %.0f solves/s, %.1f ms per step, backward kernel %.1f ms, forward %.1f ms' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['forward_kernel_ms']))"
done
