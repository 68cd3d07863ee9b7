#!/bin/bash
# Round artefacts for profiles/ in one gpurun call:  bash tools/gpu_round.sh r05
tag=${1:-r06}
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -1
{ echo "# SA_KERNEL_DEFINES=-DSA_WAVE_PROFILE python tools/profile_wave.py 1024   (network100 section timers; then -DSA_WAVE_PROFILE_PHASES; $tag, MI355X)";
  SA_KERNEL_DEFINES="-DSA_WAVE_PROFILE" timeout 600 python tools/profile_wave.py 1024 2>&1 | tail -5;
  SA_KERNEL_DEFINES="-DSA_WAVE_PROFILE -DSA_WAVE_PROFILE_PHASES" timeout 600 python tools/profile_wave.py 1024 2>&1 | tail -3;
  echo "# the same with the LU's inner timers (-DSA_LU_PROFILE_SEGMENTS), then the publication timeline (-DSA_LU_PROFILE_TIMELINE)";
  SA_KERNEL_DEFINES="-DSA_WAVE_PROFILE -DSA_LU_PROFILE_SEGMENTS" timeout 600 python tools/profile_wave.py 1024 2>&1 | tail -3;
  SA_KERNEL_DEFINES="-DSA_WAVE_PROFILE -DSA_LU_PROFILE_TIMELINE" timeout 600 python tools/profile_wave.py 1024 2>&1 | tail -3; } > gpurun_out/${tag}_network100_sections.txt 2>&1
tail -16 gpurun_out/${tag}_network100_sections.txt
timeout 900 python tools/bench_small_batch.py > gpurun_out/${tag}_small_batch.json 2> gpurun_out/${tag}_small_batch.log; tail -c 600 gpurun_out/${tag}_small_batch.json
timeout 600 python bench.py --gpus 2 --single-process --devices 0,0 --steps 5 --warmup 2 --no-cpu-baseline --no-extra-configs > gpurun_out/${tag}_single_process_two_handles.json 2>&1; tail -c 400 gpurun_out/${tag}_single_process_two_handles.json
{ for w in seir network100; do timeout 900 python bench.py --workload $w --gpus 8 --single-process --devices 0,0,0,0,0,0,0,0 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1; done; } > gpurun_out/${tag}_single_process_eight_handles.json
{ echo "# SA_KERNEL_DEFINES=-DSA_WAVE_PROFILE python tools/profile_wave.py 16384 seir   (section timers), then -DSA_WAVE_PROFILE_PHASES ($tag, MI355X)";
  SA_GUARD=0 SA_KERNEL_DEFINES="-DSA_WAVE_PROFILE" timeout 600 python tools/profile_wave.py 16384 seir 2>&1 | tail -3;
  SA_GUARD=0 SA_KERNEL_DEFINES="-DSA_WAVE_PROFILE -DSA_WAVE_PROFILE_PHASES" timeout 600 python tools/profile_wave.py 16384 seir 2>&1 | tail -3; } > gpurun_out/${tag}_seir_sections.txt
timeout 1500 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.log; tail -c 1500 gpurun_out/${tag}_bench.json
{ echo "# SA_KERNEL_DEFINES=-DSA_ABLATE_PROFILE python tools/profile_lv.py 65536 lv ; ... 262144 robertson   ($tag, MI355X; phase shares of the one-lane backward kernel)";
  SA_GUARD=0 SA_KERNEL_DEFINES="-DSA_ABLATE_PROFILE" timeout 600 python tools/profile_lv.py 65536 lv 2>&1 | tail -3;
  SA_GUARD=0 SA_KERNEL_DEFINES="-DSA_ABLATE_PROFILE" timeout 900 python tools/profile_lv.py 262144 robertson 2>&1 | tail -3; } > gpurun_out/${tag}_lv_phases.txt
tail -6 gpurun_out/${tag}_lv_phases.txt
