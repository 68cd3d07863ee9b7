"""Stand-alone bundle of the SIOptimizeVGPRLiveRange miscompile (profiles/r04_sens_anomaly.txt) for a compiler bug report.

    python tools/make_liverange_repro.py [outdir]          (CPU box: writes the bundle, default gpurun_out/liverange_repro)
    python tools/make_liverange_repro.py --run a.hsaco     (MI355X box: runs SEIR's sensitivity solve on THAT code object
                                                            and prints how many step counters differ from the CPU oracle)

The bundle holds ONE LLVM IR file -- SEIR's 4-lane forward-sensitivity kernels after this repository's front half of
the build (clang -O0, always-inline, sroa), coefficient vectors parked in LDS (-DSA_SENS_CTL_PARK) -- and the two
back-end command lines that differ in exactly one flag:

    clang -x ir k.ll -target amdgcn-amd-amdhsa -mcpu=gfx950 -O3 -ffp-contract=off -mllvm -disable-machine-licm -c -o bad.o
    clang -x ir k.ll ...                                    ... -mllvm -amdgpu-opt-vgpr-liverange=0           -c -o good.o
    ld.lld -shared bad.o -o bad.hsaco ; ld.lld -shared good.o -o good.hsaco

`bad.hsaco` integrates 21 SEIR instances with wrong step counters (129 of 273 entries differ from the oracle, up to 57
steps; the machine verifier is silent), `good.hsaco` is bit-identical to the oracle.  Not reduced further: the kernel
has ~1 800 spill slots and the failure needs the register pressure (it disappears at -O1 and when the parked values
are read once more).  Since round 5 the product itself notices such a pair (tests/test_guard.py).
"""
import lzma
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BAD_ENV = {"SA_KERNEL_DEFINES": "-DSA_SENS_CTL_PARK -DSA_SENS_UNROLL", "SA_CLANG_FLAGS": "-mllvm -disable-machine-licm", "SA_GUARD": "0"}


def make(outdir):
    import glob
    import subprocess
    os.environ.update(BAD_ENV)
    from sunode_amd import _native
    from tools.problem_cache import make_problem
    src = make_problem("seir").native_source()
    before = set(glob.glob(os.path.join(_native._CACHE, "sa_build_*")))
    _native.build_code_object(src, sens=True, force=True, keep_temps=True)
    tmp = (set(glob.glob(os.path.join(_native._CACHE, "sa_build_*"))) - before).pop()
    os.makedirs(outdir, exist_ok=True)
    ll = os.path.join(outdir, "k.ll")
    subprocess.run([os.path.join(_native.LLVM_BIN, "llvm-dis"), os.path.join(tmp, "k1.bc"), "-o", ll], check=True)
    shutil.rmtree(tmp, ignore_errors=True)
    with open(ll, "rb") as fh, lzma.open(ll + ".xz", "wb") as out:
        shutil.copyfileobj(fh, out)
    size = os.path.getsize(ll)
    os.remove(ll)
    base = "clang -x ir k.ll -target amdgcn-amd-amdhsa -mcpu=gfx950 -O3 -ffp-contract=off -mllvm -disable-machine-licm"
    with open(os.path.join(outdir, "README.txt"), "w") as fh:
        fh.write(__doc__ + "\ntoolchain: %s\nxz -d k.ll.xz    (%d bytes of IR)\n%s -c -o bad.o\n%s -mllvm "
                 "-amdgpu-opt-vgpr-liverange=0 -c -o good.o\nld.lld -shared bad.o -o bad.hsaco\nld.lld -shared good.o -o "
                 "good.hsaco\npython tools/make_liverange_repro.py --run bad.hsaco     # in this repository, on gfx950\n"
                 % (_native.toolchain_id()["banner"], size, base, base))
    print("wrote", outdir, os.listdir(outdir))


def run(path):
    import numpy as np
    os.environ["SA_GUARD"] = "0"
    from sunode_amd import _native
    from sunode_amd.solver import Solver
    from tests.helpers import make_oracle
    from tools.problem_cache import make_problem
    from tools.problems import seir_batch
    real = _native.build_code_object
    _native.build_code_object = lambda *a, **k: os.path.abspath(path) if k.get("sens") else real(*a, **k)
    prob = make_problem("seir")
    d = seir_batch(21)
    tv = d["tvals"][::5]
    sens0 = np.zeros((prob.n_params, prob.n_states)); sens0[0, 3] = 0.5
    sol = Solver(prob, abstol=1e-8, reltol=1e-8, sens_mode="simultaneous")
    y, S, st, stats = sol.solve_sens_batch(0.0, tv, d["y0"], d["ps"], d["pr"], sens0)
    orc = make_oracle("seir")
    yo, So, so, sto = orc.solve_sens(orc.config(rtol=1e-8, atol=1e-8), d["y0"], d["ps"], d["pr"], sens0, 0.0, tv,
                                     mode="simultaneous", nthreads=8)
    cmp = [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13]
    bad = int((stats[:, cmp] != sto[:, cmp]).sum())
    print("%s: %d of %d step counters differ from the oracle (largest difference %d steps); states %s"
          % (path, bad, stats[:, cmp].size, int(np.abs(stats[:, cmp] - sto[:, cmp]).max()),
             "identical" if np.array_equal(y, yo) and np.array_equal(S, So) else "DIFFER"))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--run":
        run(sys.argv[2])
    else:
        make(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "liverange_repro"))
