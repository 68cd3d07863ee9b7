#!/bin/bash
# One lane per instance vs 4-lane groups as a function of the batch size, for the model sizes around the selection boundary
# (kernel_variant: n <= 5 and p <= 8 -> one lane).   bash tools/ab_mapping_by_batch.sh > gpurun_out/r06_mapping_by_batch.txt
echo "# python tools/bench_problem.py <name> <B>: SA_FORCE_GROUP=1 | wave4 (r06, MI355X)"
for name in rn3_3 rn4_2 rn4_8 rn5_2 rn5_8 forcing; do
  for B in 64 1024 4096 16384 32768 65536; do
    for g in 1 wave4; do
      echo -n "[$g] "; SA_GUARD=0 SA_FORCE_GROUP=$g timeout 900 python tools/bench_problem.py $name $B 2>&1 | tail -1
    done
  done
done
