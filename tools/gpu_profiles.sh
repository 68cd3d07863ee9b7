#!/bin/bash
# rocprofv3 evidence for profiles/ (one gpurun call): for EVERY BASELINE configuration the kernel-trace stats of the
# bench command and separate PMC passes (MI355X_MICROARCH.md: counters in their own runs) -- HBM bytes, instruction
# mix, wave / active / wait cycles, lane utilisation inputs -- plus the FETCH_SIZE / WRITE_SIZE calibration.
#     usage: tools/gpu_profiles.sh <tag>          -> gpurun_out/<tag>_*.txt   (then: python tools/make_pmc_traffic.py <tag>)
tag=${1:-r06}
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$root"
export TMPDIR=/tmp
out="/tmp/prof_${tag}"
mkdir -p "$out" "$root/gpurun_out"
summ() { db=$(ls -t "$1"/*/*_results.db 2>/dev/null | head -1); [ -z "$db" ] && db=$(ls -t "$1"/*results.db | head -1); python tools/rocpd_summary.py "$db" "${@:2}"; }
for w in ${WORKLOADS:-lv robertson seir network100}; do
    steps=5; [ "$w" != lv ] && steps=3
    cmd="python bench.py --workload $w --steps $steps --warmup 2 --no-cpu-baseline --no-extra-configs"
    # once without the profiler: a fresh cache has no guard verdict yet, and the guard's 64-instance shadow launches
    # would otherwise sit in the kernel statistics of the profiled command
    timeout 900 $cmd > /dev/null 2>&1
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$out/kt_$w" -o $w -- bash -c "cd $root && $cmd" > "$out/kt_$w.log" 2>&1)
    { echo "# rocprofv3 --kernel-trace --stats -- $cmd   ($tag, MI355X)"; summ "$out/kt_$w"; } > "$root/gpurun_out/${tag}_${w}_kernel_stats.txt"
    grep '^{"metric"' "$out/kt_$w.log" | tail -1 > "$root/gpurun_out/${tag}_${w}_bench.json"     # the context of the counters (make_pmc_traffic.py)
    head -4 "$root/gpurun_out/${tag}_${w}_kernel_stats.txt"
    f="$root/gpurun_out/${tag}_${w}_pmc.txt"
    echo "# rocprofv3 --kernel-trace --pmc <counters> (one pass per line) -- $cmd   (per-dispatch means; KiB for *_SIZE)" > "$f"
    for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"; do
        d="$out/pmc_${w}_$(echo $pmc | tr ' ' '_' | cut -c1-40)"
        (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $pmc -d "$d" -o $w -- bash -c "cd $root && $cmd" > "$d.log" 2>&1)
        { echo "## --pmc $pmc"; summ "$d" --counters | grep -v "at::native\|rocclr"; } >> "$f"
    done
    tail -22 "$f" | head -12
done
{ echo "# FETCH_SIZE / WRITE_SIZE calibration: tools/calib_fetch.hip, 2^31 bytes per dispatch (reads) / 2^31 (write), KiB per dispatch reported"; } > "$root/gpurun_out/${tag}_pmc_calibration.txt"
if [ -x "$root/tools/calib_fetch.bin" ]; then
for pmc in FETCH_SIZE WRITE_SIZE; do
    d="$out/calib_$pmc"
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d "$d" -o calib -- "$root/tools/calib_fetch.bin" > "$d.log" 2>&1)
    { echo "## --pmc $pmc"; summ "$d" --counters; } >> "$root/gpurun_out/${tag}_pmc_calibration.txt"
done
cat "$root/gpurun_out/${tag}_pmc_calibration.txt"
fi
rm -rf "$out"
