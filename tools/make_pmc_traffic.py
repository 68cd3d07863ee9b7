"""profiles/pmc_traffic.json from the PMC summaries of tools/gpu_profiles.sh (no hand-copied numbers).

    python tools/make_pmc_traffic.py r03        reads profiles/r03_<workload>_pmc.txt (or gpurun_out/), writes profiles/pmc_traffic.json

Units after conversion: bytes per launch (fetch: FETCH_SIZE KiB x 1024 x 2 -- the counter reports half the bytes read on
gfx950, profiles/r02_pmc_calibration.txt; write: WRITE_SIZE KiB x 1024); fp64 lane-operations per launch
(2 FMA + MUL + ADD wave-instructions x 64); SQ cycle counters as reported (units of four clocks, summed over waves)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r"(sa_k_\w+)\s+(\w+)\s+(\d+)\s+([0-9.e+]+)", line)
        if m:
            out.setdefault(m.group(1), {})[m.group(2)] = float(m.group(4))
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    res = {"source": "profiles/%s_<workload>_pmc.txt (rocprofv3 --kernel-trace --pmc, one pass per counter group, of "
                     "`python bench.py --workload <w> --steps N --warmup 2 --no-cpu-baseline --no-extra-configs`)" % tag,
           "unit": "bytes per launch (fetch, write); fp64 lane-operations per launch (valu_fp64_flops); SQ counters raw",
           "calibration": "FETCH_SIZE x2 (reports half the bytes read on gfx950), WRITE_SIZE x1, KiB units: "
                          "profiles/r02_pmc_calibration.txt / %s_pmc_calibration.txt" % tag}
    for w in ("lv", "robertson", "seir", "network100"):
        path = next((p for p in (os.path.join(ROOT, d, "%s_%s_pmc.txt" % (tag, w)) for d in ("profiles", "gpurun_out"))
                     if os.path.exists(p)), None)
        if not path:
            continue
        res[w] = {}
        # what the counters belong to: the bench line of the kernel-trace run of the same command (tools/gpu_profiles.sh
        # keeps it): batch, record size, code-object hash, toolchain, kernel times -- bench.py prints the counter ratios
        # only when they match the run at hand
        bpath = next((p for p in (os.path.join(ROOT, d, "%s_%s_bench.json" % (tag, w)) for d in ("profiles", "gpurun_out"))
                      if os.path.exists(p)), None)
        if bpath:
            line = json.load(open(bpath))
            b = line.get("build", {})
            res[w]["context"] = {"batch": b.get("batch"), "record_bytes": b.get("record_bytes"),
                                 "code_object": b.get("code_object"), "toolchain": (b.get("toolchain") or {}).get("hash"),
                                 "kernel_ms": {"sa_k_backward": line["roofline"]["kernel_ms"],
                                               "sa_k_forward": line["roofline"]["forward_kernel_ms"]},
                                 "source": os.path.relpath(bpath, ROOT)}
        for k, c in parse(path).items():
            e = {}
            if "FETCH_SIZE" in c:
                e["fetch"] = c["FETCH_SIZE"] * 1024 * 2
            if "WRITE_SIZE" in c:
                e["write"] = c["WRITE_SIZE"] * 1024
            if "SQ_INSTS_VALU_FMA_F64" in c:
                e["valu_fp64_flops"] = 64 * (2 * c["SQ_INSTS_VALU_FMA_F64"] + c.get("SQ_INSTS_VALU_MUL_F64", 0)
                                             + c.get("SQ_INSTS_VALU_ADD_F64", 0))
            for src, dst in (("SQ_INSTS_VALU", "valu_insts"), ("SQ_INSTS_SALU", "salu_insts"),
                             ("SQ_WAVE_CYCLES", "wave_cycles"), ("SQ_ACTIVE_INST_ANY", "active_inst_any"),
                             ("SQ_WAIT_ANY", "wait_any"), ("SQ_ACTIVE_INST_VALU", "active_inst_valu")):
                if src in c:
                    e[dst] = c[src]
            if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
                e["lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"])
            if c.get("SQ_WAVE_CYCLES") and c.get("SQ_ACTIVE_INST_VALU"):
                e["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
            res[w][k] = e
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({w: {k: {a: (("%.3g" % b) if isinstance(b, float) else b) for a, b in v.items()} for k, v in res[w].items()}
                      for w in res if isinstance(res[w], dict)}, indent=1))


if __name__ == "__main__":
    main()
