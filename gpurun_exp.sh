mkdir -p gpurun_out
(time python -m pytest tests -m gpu -q --timeout 1500 --durations=8 2>&1 | tail -20) > gpurun_out/r05_gputests.log 2>&1; tail -6 gpurun_out/r05_gputests.log
WORKLOADS=seir bash tools/gpu_profiles.sh r05 2>&1 | tail -6
{ echo "# SA_KERNEL_DEFINES=-DSA_WAVE_PROFILE python tools/profile_wave.py 16384 seir   (section timers), then -DSA_WAVE_PROFILE_PHASES (r05, MI355X; callbacks as lane families)";
  SA_GUARD=0 SA_KERNEL_DEFINES="-DSA_WAVE_PROFILE" python tools/profile_wave.py 16384 seir 2>&1 | tail -3;
  SA_GUARD=0 SA_KERNEL_DEFINES="-DSA_WAVE_PROFILE -DSA_WAVE_PROFILE_PHASES" python tools/profile_wave.py 16384 seir 2>&1 | tail -3; } > gpurun_out/r05_seir_sections.txt
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.log; python -c "
import json; d=json.loads(open('gpurun_out/r05_bench.json').read().strip().splitlines()[-1]); print('lv', round(d['value']), d['ms_per_step'], d['roofline']['traffic_over_algorithmic'], d['roofline']['valu']['valu_insts_per_attempt']); print({k:(round(v.get('solves_per_s',0)), v.get('traffic_over_algorithmic')) for k,v in d.get('configs',{}).items()})"
