#!/bin/bash
# Lanes per instance as a function of the batch size for the mid-size models (engine's choice: 4 lanes up to 16 states).
#   bash tools/ab_lanes_by_batch.sh > gpurun_out/r06_lanes_by_batch.txt
echo "# python tools/bench_problem.py <name> <B>: default (4 lanes) | SA_FORCE_GROUP=wave8 | wave16 (r06, MI355X)"
for name in seir rn12_4 rn7_4; do
  for B in 64 1024 4096 16384; do
    for g in "" wave8 wave16; do
      echo -n "[${g:-default}] "; SA_GUARD=0 SA_FORCE_GROUP=$g timeout 900 python tools/bench_problem.py $name $B 2>&1 | tail -1
    done
  done
done
