#!/bin/bash
# Where should the one-lane-per-instance (register) kernel hand over to the 4-lane groups?  kernel_variant: n <= 5 and p <= 8.
#   bash tools/ab_mapping_boundary.sh > gpurun_out/r06_mapping_boundary.txt
echo "# python tools/bench_problem.py <name> 16384: default mapping | SA_FORCE_GROUP=wave4 | SA_FORCE_GROUP=1 (r06, MI355X)"
for name in forcing rn5_8 rn5_9 rn6_1 lv12 misc; do
  for g in "" wave4 1; do
    echo -n "[${g:-default}] "; SA_GUARD=0 SA_FORCE_GROUP=$g timeout 900 python tools/bench_problem.py $name 16384 2>&1 | tail -1
  done
done
