cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=12 -x -q > gpurun_out/r03a_tests.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r03a_tests.log
