cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== G=8 lean+cold"; timeout 600 python tools/bench_configs.py seir 16384 2>&1 | tail -1
echo "== G=4 lean+cold"; SA_FORCE_GROUP=wave4 timeout 600 python tools/bench_configs.py seir 16384 2>&1 | tail -1
echo "== G=4 sections"; SA_FORCE_GROUP=wave4 SA_KERNEL_DEFINES=-DSA_WAVE_PROFILE timeout 400 python tools/profile_wave.py 16384 seir 2>&1 | tail -3
echo "== G=4 phases"; SA_FORCE_GROUP=wave4 SA_KERNEL_DEFINES="-DSA_WAVE_PROFILE -DSA_WAVE_PROFILE_PHASES" timeout 400 python tools/profile_wave.py 16384 seir 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "seir or row_exchanges or edge_cases" --maxfail=6 > gpurun_out/r03b_tests.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/r03b_tests.log
