#!/bin/bash
# One gpurun call of the round: new kernels first (short timeout), then the suite, the bench line, profiles.
# usage: tools/gpu_round.sh <tag>          (writes gpurun_out/<tag>_*)
tag=${1:-r02}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== new kernels"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider \
    -k "matvec or network100 or row_exchanges" --maxfail=4 > gpurun_out/${tag}_new.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/${tag}_new.log
echo "== suite"; timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=12 > gpurun_out/${tag}_tests.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/${tag}_tests.log
echo "== bench"; timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "rc=$?"; tail -c 600 gpurun_out/${tag}_bench.err
echo "== network100 section timers"; SA_KERNEL_DEFINES=-DSA_WAVE_PROFILE timeout 400 python tools/profile_wave.py 1024 > gpurun_out/${tag}_net100_profile.txt 2>&1; cat gpurun_out/${tag}_net100_profile.txt | tail -5
echo "== seir section timers"; SA_KERNEL_DEFINES=-DSA_WAVE_PROFILE timeout 400 python tools/profile_wave.py 16384 seir > gpurun_out/${tag}_seir_profile.txt 2>&1; tail -4 gpurun_out/${tag}_seir_profile.txt
echo "== phase timers"; for w in "1024" "16384 seir"; do SA_KERNEL_DEFINES="-DSA_WAVE_PROFILE -DSA_WAVE_PROFILE_PHASES" timeout 400 python tools/profile_wave.py $w 2>&1 | tail -3; done > gpurun_out/${tag}_phases.txt; cat gpurun_out/${tag}_phases.txt
echo "== small batches"; timeout 600 python tools/bench_small_batch.py 1e-10 > gpurun_out/${tag}_small_batch.json 2>&1; python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_small_batch.json"))
for k in ("thread_per_instance","cooperative_8_lanes"):
    print(k, [(r["B"], round(r["wall_ms"],2), round(r["kernel_ms"],2)) for r in d[k]])
PY
