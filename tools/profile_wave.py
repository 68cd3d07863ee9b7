"""Section timers of the wavefront-per-instance kernel (SA_KERNEL_DEFINES=-DSA_WAVE_PROFILE builds).

python tools/profile_wave.py <B> [generated-header | seir]     (10 ns ticks from stats slots 9..15)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sunode_amd import SympyProblem  # noqa: E402
from sunode_amd.solver import AdjointSolver  # noqa: E402
from tools.problems import network100, network_batch  # noqa: E402


def main():
    B = int(sys.argv[1])
    if len(sys.argv) > 2 and sys.argv[2] == "seir":
        from tools.problems import PROBLEMS, seir_batch
        s = PROBLEMS["seir"]
        prob = SympyProblem(s["params"], s["states"], s["rhs"], s["derivative_params"])
        d = seir_batch(B)
    else:
        s = network100()
        prob = SympyProblem(s["params"], s["states"], s["rhs"], s["derivative_params"])
        if len(sys.argv) > 2:
            prob._native_source = open(sys.argv[2]).read()
        d = network_batch(B)
    tv = d["tvals"]
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(prob.n_states)[None, :])
    sol = AdjointSolver(prob, abstol=1e-8, reltol=1e-8, backward_abstol=1e-8, backward_reltol=1e-8,
                        quad_abstol=1e-8, quad_reltol=1e-8, max_steps=1024)
    for rep in range(2):
        y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], d["ps"], d["pr"])
        g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    f, b = sol._engine().last_kernel_ms()
    print("B=%d fwd %.1f ms bwd %.1f ms -> %.0f solves/s" % (B, f, b, B / ((f + b) * 1e-3)))
    names = ["rhs", "quad", "jac", "getrf", "getrs", "copy"]
    if "PROFILE_PHASES" in os.environ.get("SA_KERNEL_DEFINES", ""):
        names = ["pre_step", "predict+set", "interp", "newton", "errtest+quad", "complete+prepare"]
    for tag, stt in (("fwd", stats), ("bwd", statsb)):
        tot = stt[:, 15].mean() * 1e-5
        parts = ", ".join("%s %.1f" % (n, stt[:, 9 + i].mean() * 1e-5) for i, n in enumerate(names))
        if tag == "bwd" and "PROFILE_PHASES" in os.environ.get("SA_KERNEL_DEFINES", "") and stt[:, 8].mean() > 1e4 \
                and len(sys.argv) > 2 and sys.argv[2] == "seir":      # (lane-group builds: slot 8 = time in restarts)
            parts += ", restarts %.1f" % (stt[:, 8].mean() * 1e-5)
        if (stt[:, 8] & 0xffffffff).sum() > 0 and (stt[:, 8] >> 32).sum() == 0 and stt[:, 2].mean() > 0 and \
                "PROFILE_PHASES" not in os.environ.get("SA_KERNEL_DEFINES", ""):      # workgroup LU: its own wall clock
            print("    LU (setup_lu_regs, wavefront 0): %.1f us per factorisation"
                  % ((stt[:, 8] & 0xffffffff).mean() / max(stt[:, 2].mean(), 1) / 1e2))
        if tag == "bwd" and (stt[:, 8] >> 32).sum() > 0 and (stt[:, 8] >> 32).max() < (1 << 30):   # workgroup-LU builds only
            nlu = max(stt[:, 2].mean(), 1)
            # the three partial counters are s_memtime reads the compiler may move across VALU work: indicative only
            print("    LU of wavefront 0, kilo-cycles per factorisation: panel factorisation %.1f, waiting %.1f, update %.1f"
                  % tuple(stt[:, 5 + i].mean() / nlu / 1e3 for i in range(3))
                  + "; whole function %.1f us = %.1f kilo-cycles" % ((stt[:, 8] & 0xffffffff).mean() / nlu / 1e2,
                                                                   (stt[:, 8] >> 32).mean() / nlu / 1e3))
        if "SA_LU_PROFILE_TIMELINE" in os.environ.get("SA_KERNEL_DEFINES", ""):
            # publication time of panel p (cycles after the first barrier), averaged over the instances that recorded p
            pan = stt[:, 14].astype(int)
            tl = [(stt[pan == q, 13] / np.maximum(stt[pan == q, 2], 1)).mean() for q in range(pan.max() + 1)]
            print("    LU timeline, kilo-cycles from the first barrier to the publication of panel p: "
                  + " ".join("%.1f" % (v / 1e3) for v in tl))
            print("    ... per panel: " + " ".join("%.1f" % ((b - a) / 1e3) for a, b in zip([0.0] + tl[:-1], tl)))
            continue
        if "SA_LU_PROFILE_SEGMENTS" in os.environ.get("SA_KERNEL_DEFINES", ""):
            nlu = max(stt[:, 2].mean(), 1)
            print("    LU segments of wavefront 0, kilo-cycles per factorisation: load+form %.1f, first barrier %.1f, "
                  "write-back %.1f, last barrier %.1f; owner block: own four columns %.1f, four steps %.1f (rest: publication)"
                  % tuple(stt[:, 9 + i].mean() / nlu / 1e3 for i in range(6)))
            continue
        print("%s per-instance ms: total %.1f | %s | nst %.0f nfe %.0f nsetups %.0f nje %.0f nni %.0f"
              % (tag, tot, parts, stt[:, 0].mean(), stt[:, 1].mean(), stt[:, 2].mean(), stt[:, 3].mean(),
                 stt[:, 4].mean()))


if __name__ == "__main__":
    main()
