#!/bin/bash
# Reproducer (round 4) for the miscompile behind the "parking l[] changes the results" anomaly of the lean 4-lane
# forward-sensitivity build (bdf_wave.hip, -DSA_SENS; ADVICE r3 / VERDICT r3 #3).  On an MI355X box, from the repo root:
#     bash tools/repro_vgpr_liverange.sh
# builds SEIR's sensitivity kernel four ways and runs the bit-exactness test against the oracle:
#   1. coefficient vectors parked in LDS + machine LICM off                       -> FAILS (half of the step counters differ)
#   2. the same + -amdgpu-opt-vgpr-liverange=0 (SIOptimizeVGPRLiveRange off)      -> passes
#   3. the same as 1 at -O1                                                        -> passes
# (round 4 also ran 1 with a uniformity diagnostic, -DSA_CTL_CHECK -- every lane compares l / tau / tq with lane 0 of its
#  group: passes, vectors uniform; reading the values once more hides the problem.  The diagnostic left the kernel file
#  in round 5; the record is profiles/r04_sens_anomaly.txt section 2.)
# Since round 5 the PRODUCT catches this combination by itself: tests/test_guard.py::test_guard_catches_the_round4_miscompile.
# The difference 1 <-> 2 survives: no VGPR->AGPR spilling, no SGPR->VGPR spilling, no machine CSE, no pre-/post-RA
# scheduler, no early if-conversion, sink splitting on/off, a hard s_barrier + s_waitcnt around the parked values, the
# machine verifier (silent).  Record: profiles/r04_sens_anomaly.txt.  The product builds pass -amdgpu-opt-vgpr-liverange=0
# (sunode_amd/_native.py SAFETY_CODEGEN_FLAGS).
T="tests/test_forward_sens.py"
export SA_GUARD=0        # (the differential guard would repair build 1)
run() { echo "=== $1 | defines: $2 | flags: $3"; SA_CLANG_FLAGS="$3" SA_KERNEL_DEFINES="$2" timeout 900 python -m pytest $T -q -m gpu \
        -k "seir_lane_groups and None" 2>&1 | grep -E "passed|failed|Mismatched" | head -4; }
F="-mllvm -disable-machine-licm"
run "1. parked, machine LICM off" "-DSA_SENS_CTL_PARK -DSA_SENS_UNROLL" "$F"
run "2. ... SIOptimizeVGPRLiveRange off" "-DSA_SENS_CTL_PARK -DSA_SENS_UNROLL" "$F -mllvm -amdgpu-opt-vgpr-liverange=0"
run "3. ... at -O1" "-DSA_SENS_CTL_PARK -DSA_SENS_UNROLL" "$F -O1"
