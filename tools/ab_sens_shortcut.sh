#!/bin/bash
# A/B: wave-uniform shortcuts (sa_common.h SA_SHORTCUT) in the forward-sensitivity builds: off (default) | -DSA_SENS_SHORTCUT
echo "# python tools/bench_sens.py <problem> <B> simultaneous, default | SA_KERNEL_DEFINES=-DSA_SENS_SHORTCUT (r06, MI355X)"
for pb in "lv 65536" "robertson 65536" "seir 16384"; do
  for defs in "" "-DSA_SENS_SHORTCUT"; do
    echo -n "[${defs:-default}] "; SA_GUARD=0 SA_KERNEL_DEFINES="$defs" timeout 900 python tools/bench_sens.py $pb simultaneous 2>&1 | tail -1
  done
done
echo "# Robertson's sensitivity solve through the lane-group kernel (workspace-streamed vectors) instead of the register kernel (1 136 spill slots): SA_FORCE_GROUP=wave<G>"
for g in wave2 wave4; do
    echo -n "[SA_FORCE_GROUP=$g] "; SA_GUARD=0 SA_FORCE_GROUP=$g timeout 900 python tools/bench_sens.py robertson 65536 simultaneous 2>&1 | tail -1
done
