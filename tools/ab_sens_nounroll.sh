#!/bin/bash
# A/B: the parameter loop of the lane-group sensitivity corrector as a loop (default since round 6) or unrolled
# (-DSA_SENS_UNROLL, the round-5 form).   bash tools/ab_sens_nounroll.sh > gpurun_out/r06_sens_nounroll.txt
echo "# python tools/bench_sens.py seir 16384 <mode>, default | SA_KERNEL_DEFINES=-DSA_SENS_UNROLL (r06, MI355X)"
for mode in simultaneous staggered; do
  for defs in "" "-DSA_SENS_UNROLL"; do
    echo -n "[${defs:-default}] "; SA_GUARD=0 SA_KERNEL_DEFINES="$defs" timeout 900 python tools/bench_sens.py seir 16384 $mode 2>&1 | tail -1
  done
done
