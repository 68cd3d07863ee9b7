#!/bin/bash
# A/B (run BEFORE the lean division became the default; the switch is -DSA_TABLE_IEEE_DIV now): IEEE divisions or the lean
# division sa_common.h fdiv (-DSA_TABLE_FDIV; same quotients bit for bit in this operand range).
#   bash tools/ab_table_fdiv.sh > gpurun_out/r06_table_fdiv.txt        (MI355X box; code objects pre-built)
echo "# python bench.py --workload <w> --steps 5 --warmup 2, default build | SA_KERNEL_DEFINES=-DSA_TABLE_FDIV (r06, MI355X)"
for w in lv robertson seir; do
  for rep in 1 2; do
    for defs in "" "-DSA_TABLE_FDIV"; do
        SA_GUARD=0 SA_KERNEL_DEFINES="$defs" timeout 900 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 | \
          python -c "import sys, json; d = json.loads(sys.stdin.read()); r = d['roofline']; print('$w [%s] run $rep: %.4g solves/s, forward %.2f ms, backward %.2f ms' % ('$defs' or 'default', d['value'], r['forward_kernel_ms'], r['kernel_ms']))"
    done
  done
done
