"""Throughput of the forward-sensitivity path: python tools/bench_sens.py lv 65536 [simultaneous|staggered]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sunode_amd import SympyProblem  # noqa: E402
from sunode_amd.solver import Solver  # noqa: E402
from tools.problems import PROBLEMS, lv_batch, robertson_batch, seir_batch  # noqa: E402


def main():
    name, B = sys.argv[1], int(sys.argv[2])
    mode = sys.argv[3] if len(sys.argv) > 3 else "simultaneous"
    s = PROBLEMS[name]
    prob = SympyProblem(s["params"], s["states"], s["rhs"], s["derivative_params"])
    if name == "lv":
        d = lv_batch(B); ps, pr = d["params"][:, :2], d["params"][:, 2:]; rt, at = 1e-8, 1e-8
    elif name == "seir":
        d = seir_batch(B); ps, pr = d["ps"], d["pr"]; rt, at = 1e-8, 1e-8
    else:
        d = robertson_batch(B); ps, pr = d["params"], np.zeros(0); rt, at = 1e-8, 1e-10
    sol = Solver(prob, abstol=at, reltol=rt, sens_mode=mode)
    sens0 = np.zeros((prob.n_params, prob.n_states))
    for rep in range(2):
        t0 = time.perf_counter()
        y, S, st, stats = sol.solve_sens_batch(0.0, d["tvals"], d["y0"], ps, pr, sens0)
        wall = time.perf_counter() - t0
    f, _ = sol._engine().last_kernel_ms()
    print("%s %s B=%d: kernel %.2f ms, wall %.1f ms -> %.3g solves/s (kernel), failed %d, steps %.0f"
          % (name, mode, B, f, 1e3 * wall, B / (f * 1e-3), int((st != 0).sum()), stats[:, 0].mean()))


if __name__ == "__main__":
    main()
