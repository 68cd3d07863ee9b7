(time python -m pytest tests -m gpu -q --timeout 1500 --durations=8 2>&1 | tail -16) > gpurun_out/r05_gputests.log 2>&1; tail -5 gpurun_out/r05_gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
