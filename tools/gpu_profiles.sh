#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of the bench command for every BASELINE configuration,
# PMC passes (separate, as MI355X_MICROARCH.md prescribes) for the headline configuration, and the FETCH_SIZE /
# WRITE_SIZE calibration on known byte counts.    usage: tools/gpu_profiles.sh <tag>
tag=${1:-r02}
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$root"
export TMPDIR=/tmp
out="$root/gpurun_out/prof_${tag}"
mkdir -p "$out"
summ() { db=$(ls -t "$1"/*/*_results.db 2>/dev/null | head -1); [ -z "$db" ] && db=$(ls -t "$1"/*results.db | head -1); python tools/rocpd_summary.py "$db" "${@:2}"; }
for w in lv robertson seir network100; do
    steps=5; [ "$w" != lv ] && steps=2
    cmd="python bench.py --workload $w --steps $steps --warmup 1 --no-cpu-baseline --no-extra-configs"
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$out/kt_$w" -o $w -- bash -c "cd $root && $cmd" > "$out/kt_$w.log" 2>&1)
    { echo "# rocprofv3 --kernel-trace --stats -- $cmd   (round 2, MI355X)"; summ "$out/kt_$w"; } > "$root/gpurun_out/${tag}_${w}_kernel_stats.txt"
    tail -1 "$out/kt_$w.log" | cut -c1-300
    head -4 "$root/gpurun_out/${tag}_${w}_kernel_stats.txt"
done
cmd="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs"
{ echo "# rocprofv3 --kernel-trace --pmc <counters> (one pass per line) -- $cmd   (round 2; per-dispatch means)"; } > "$root/gpurun_out/${tag}_lv_pmc.txt"
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"; do
    d="$out/pmc_$(echo $pmc | tr ' ' '_' | cut -c1-40)"
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $pmc -d "$d" -o lv -- bash -c "cd $root && $cmd" > "$d.log" 2>&1)
    { echo "## --pmc $pmc"; summ "$d" --counters | grep -v "at::native\|rocclr"; } >> "$root/gpurun_out/${tag}_lv_pmc.txt"
done
{ echo "# FETCH_SIZE / WRITE_SIZE calibration: tools/calib_fetch.hip, 2^31 bytes per dispatch (reads) / 2^31 (write), KiB per dispatch reported"; } > "$root/gpurun_out/${tag}_pmc_calibration.txt"
for pmc in FETCH_SIZE WRITE_SIZE; do
    d="$out/calib_$pmc"
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d "$d" -o calib -- "$root/tools/calib_fetch.bin" > "$d.log" 2>&1)
    { echo "## --pmc $pmc"; summ "$d" --counters; } >> "$root/gpurun_out/${tag}_pmc_calibration.txt"
done
cat "$root/gpurun_out/${tag}_pmc_calibration.txt" "$root/gpurun_out/${tag}_lv_pmc.txt"
rm -rf "$out"      # raw rocprofv3 databases: the summaries above are what is kept (gpurun copies back at most 64 MiB)
