#!/bin/bash
# one rocprofv3 --pmc pass per counter group for one bench workload: tools/gpu_pmc_one.sh <tag> <workload> "<counters>" ["<counters>" ...]
tag=$1; w=$2; shift 2
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$root"
export TMPDIR=/tmp
out="$root/gpurun_out/prof_${tag}"
mkdir -p "$out"
summ() { db=$(ls -t "$1"/*/*_results.db 2>/dev/null | head -1); [ -z "$db" ] && db=$(ls -t "$1"/*results.db | head -1); python tools/rocpd_summary.py "$db" "${@:2}"; }
cmd="python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs"
f="$root/gpurun_out/${tag}_${w}_pmc.txt"
echo "# rocprofv3 --kernel-trace --pmc <counters> (one pass per line) -- $cmd   (per-dispatch means)" > "$f"
for pmc in "$@"; do
    d="$out/pmc_${w}_$(echo $pmc | tr ' ' '_' | cut -c1-40)"
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $pmc -d "$d" -o $w -- bash -c "cd $root && $cmd" > "$d.log" 2>&1)
    { echo "## --pmc $pmc"; summ "$d" --counters | grep -v "at::native\|rocclr"; } >> "$f"
done
cat "$f"
rm -rf "$out"
