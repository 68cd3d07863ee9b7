mkdir -p gpurun_out
{ echo "# tools/repro/liverange (k.ll + the two clang command lines of README.txt) -> bad.hsaco / good.hsaco, run through python tools/make_liverange_repro.py --run (r05, MI355X)";
  python tools/make_liverange_repro.py --run gpurun_tmp/bad.hsaco 2>&1 | tail -1; python tools/make_liverange_repro.py --run gpurun_tmp/good.hsaco 2>&1 | tail -1; } > gpurun_out/r05_liverange_repro.txt
cat gpurun_out/r05_liverange_repro.txt
(time python -m pytest tests -m gpu -q --timeout 1500 --durations=12 2>&1 | tail -25) > gpurun_out/r05_gputests.log 2>&1; tail -8 gpurun_out/r05_gputests.log
bash tools/gpu_round.sh r05 2>&1 | tail -30
{ echo "# SA_KERNEL_DEFINES=-DSA_WAVE_PROFILE python tools/profile_wave.py 16384 seir   (section timers), then -DSA_WAVE_PROFILE_PHASES (r05, MI355X)";
  SA_GUARD=0 SA_KERNEL_DEFINES="-DSA_WAVE_PROFILE" python tools/profile_wave.py 16384 seir 2>&1 | tail -3;
  SA_GUARD=0 SA_KERNEL_DEFINES="-DSA_WAVE_PROFILE -DSA_WAVE_PROFILE_PHASES" python tools/profile_wave.py 16384 seir 2>&1 | tail -3; } > gpurun_out/r05_seir_sections.txt
{ for w in seir network100; do python bench.py --workload $w --gpus 8 --single-process --devices 0,0,0,0,0,0,0,0 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1; done; } > gpurun_out/r05_single_process_eight_handles.json
python -c "
import json
for l in open('gpurun_out/r05_single_process_eight_handles.json'):
    d=json.loads(l); print(d['config']['workload'][:40], d['n_gpus'], round(d['value']), d['ms_per_step'], d.get('rank_ms_per_step'))"
bash tools/gpu_profiles.sh r05 2>&1 | tail -40
