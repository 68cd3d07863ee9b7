"""Small-batch regime of the Lotka-Volterra forward+adjoint solve (BASELINE config 1/2 problem).

The reference's PyMC use is ONE draw per call: 1.25 ms for forward + adjoint at the default 1e-10 tolerances
(/root/reference/notebooks/from_sympy.ipynb:172, one CPU core, CVODES).  This prints latency and throughput of
the engine at B = 1, 8, 64, 1024, 8192, 65536 through the host-array batch API (what a sampler calling
solve_forward / solve_backward pays, PCIe copies included) for the mappings that can carry the problem:
thread-per-instance (the engine's choice for n <= 5) and the cooperative 8-lanes-per-instance build.

python tools/bench_small_batch.py [tol]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sunode_amd import SympyProblem  # noqa: E402
from tools.problems import PROBLEMS, lv_batch  # noqa: E402


def run(variant, tol, sizes):
    if variant:
        os.environ["SA_FORCE_GROUP"] = variant
    else:
        os.environ.pop("SA_FORCE_GROUP", None)
    from sunode_amd.solver import AdjointSolver
    s = PROBLEMS["lv"]
    prob = SympyProblem(s["params"], s["states"], s["rhs"], s["derivative_params"])
    sol = AdjointSolver(prob, abstol=tol, reltol=tol, backward_abstol=tol, backward_reltol=tol, quad_abstol=tol,
                        quad_reltol=tol)
    rows = []
    for B in sizes:
        d = lv_batch(B)
        ps = d["params"][:, prob.params_subset.subset_index]
        pr = d["params"][:, prob.params_subset.remainder_index]
        tv = d["tvals"]
        g = np.ones((len(tv), 2))
        reps = 3 if B >= 8192 else 10
        best = 1e9
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            y, st, _ = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
            go, lam, stb, _ = sol.solve_backward_batch(tv[-1], 0.0, tv, g)
            dt = time.perf_counter() - t0
            best = min(best, dt)
        f, b = sol._engine().last_kernel_ms()
        assert (st == 0).all() and (stb == 0).all()
        rows.append(dict(B=B, wall_ms=1e3 * best, kernel_ms=f + b, solves_per_s=B / best))
    sol._engine().close()
    return rows


def main():
    tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-10
    sizes = [1, 8, 64, 1024, 8192, 65536]
    out = {"problem": "Lotka-Volterra forward+adjoint, 2 sens params, n_t=50", "tol": tol,
           "reference_one_draw_ms": 1.25, "reference_source": "notebooks/from_sympy.ipynb:172 (CVODES, 1 CPU core)"}
    for name, variant in (("thread_per_instance", None), ("cooperative_8_lanes", "8")):
        out[name] = run(variant, tol, sizes if variant is None else sizes[:4])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
