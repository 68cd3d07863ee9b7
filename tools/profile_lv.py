"""Phase timers of the thread-per-instance kernel (SA_KERNEL_DEFINES=-DSA_ABLATE_PROFILE builds; results of such a
build are NOT bit-exact material, timing only).

SA_KERNEL_DEFINES=-DSA_ABLATE_PROFILE python tools/profile_lv.py [B] [lv|robertson]
Each lane charges the clock ticks between two phase marks to the phase it was in, so a lane idling while its
neighbours run a divergent block charges that time to its own current phase.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sunode_amd import SympyProblem  # noqa: E402
from sunode_amd.solver import AdjointSolver  # noqa: E402
from tools.problems import PROBLEMS, lv_batch, robertson_batch  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    name = sys.argv[2] if len(sys.argv) > 2 else "lv"
    s = PROBLEMS[name]
    prob = SympyProblem(s["params"], s["states"], s["rhs"], s["derivative_params"])
    if name == "lv":
        d = lv_batch(B); ps, pr = d["params"][:, :2], d["params"][:, 2:]; at = 1e-8
    else:
        d = robertson_batch(B); ps, pr = d["params"], np.zeros(0); at = 1e-10
    tv = d["tvals"]
    grads = np.ones((len(tv), prob.n_states))
    if name != "lv":       # (Robertson conserves y1 + y2 + y3: a cotangent of ones has a zero adjoint)
        grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(prob.n_states)[None, :])
    sol = AdjointSolver(prob, abstol=at, reltol=1e-8, backward_abstol=at, backward_reltol=1e-8,
                        quad_abstol=at, quad_reltol=1e-8)
    for rep in range(2):
        y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
        g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    f, b = sol._engine().last_kernel_ms()
    print("%s B=%d fwd %.2f ms bwd %.2f ms -> %.3g solves/s" % (name, B, f, b, B / ((f + b) * 1e-3)))
    # marks: kernel loop top (0), bdf_core.h cv_attempt PH_ADD 1..5, end of an observation interval (7)
    names = ["step tail + output checks + rejected attempts", "pre_step + adjust + predict + cvSet", "interpolation",
             "newton", "error test + quadrature", "complete + prepare", "(unused)",
             "waiting for the slowest lane of the wavefront at the end of an observation interval + restart"]
    if "SA_SEARCH_COUNT" in os.environ.get("SA_KERNEL_DEFINES", ""):
        print("index search, dependent global loads per instance: moves to the left %.1f, to the right %.1f (of %.0f attempts)"
              % (statsb[:, 11].mean(), statsb[:, 12].mean(), statsb[:, 14].mean()))
        return
    if "SA_INTERP_PROFILE" in os.environ.get("SA_KERNEL_DEFINES", ""):
        names = ["everything but the interpolation", "index search", "point loads of a rebuild (until they have arrived)",
                 "table arithmetic", "table stored + touches issued", "evaluation (table reads + Horner)", "(unused)",
                 names[7]]
    p = statsb[:, 8:16].astype(float)
    tot = p.sum(axis=1).mean()
    print("bwd: steps %.0f, nfe %.0f, nsetups %.0f, nni %.0f, netf %.0f"
          % (statsb[:, 0].mean(), statsb[:, 1].mean(), statsb[:, 2].mean(), statsb[:, 4].mean(), statsb[:, 6].mean()))
    print("bwd phase shares: " + ", ".join("%s %.1f%%" % (n, 100 * p[:, i].mean() / tot) for i, n in enumerate(names)))


if __name__ == "__main__":
    main()
