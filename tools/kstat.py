"""Register / LDS / scratch footprint of a problem's code object (llvm-readelf --notes on the .hsaco).

python tools/kstat.py seir [--sens] [--hermite]        (honours SA_FORCE_GROUP / SA_KERNEL_DEFINES / SA_WAVES_PER_EU)
"""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sunode_amd import SympyProblem, _native  # noqa: E402


def kernel_stats(path):
    txt = subprocess.run([os.path.join(_native.LLVM_BIN, "llvm-readelf"), "--notes", path],
                         capture_output=True, text=True).stdout
    out = []
    for blk in txt.split("- .agpr_count:")[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        out.append((g("name"), g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"),
                    g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    return out


def main():
    from tools.problems import EXTRA_PROBLEMS, PROBLEMS, network100
    name = sys.argv[1]
    s = network100() if name == "network100" else {**PROBLEMS, **EXTRA_PROBLEMS}[name]
    prob = SympyProblem(s["params"], s["states"], s["rhs"], s["derivative_params"])
    t0 = time.time()
    path = _native.build_code_object(prob.native_source(), sens="--sens" in sys.argv, hermite="--hermite" in sys.argv,
                                     constraints="--constraints" in sys.argv, force="--force" in sys.argv)
    print("%s (%.0f s, %d bytes)" % (path, time.time() - t0, os.path.getsize(path)))
    print("%-14s %5s %5s %5s %6s %6s %8s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "lds"))
    for row in kernel_stats(path):
        print("%-14s %5s %5s %5s %6s %6s %8s %6s" % row)


if __name__ == "__main__":
    main()
