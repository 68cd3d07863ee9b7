"""Forward + adjoint kernel time of any named problem (tools/problem_cache.py) on a synthetic batch:
    python tools/bench_problem.py <name> <B>            (SA_FORCE_GROUP / SA_KERNEL_DEFINES select the mapping / build)
Used for mapping-boundary measurements (profiles/r06_mapping_boundary.txt)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def batch(name, B):
    from tools import problems as P
    from tools.sweep_cases import batch_of
    named = {"forcing": P.forcing_batch, "logistic_switch": P.logistic_switch_batch, "misc": P.misc_batch}
    if name in named:
        return named[name](B)
    if name in ("seir", "robertson", "robertson5"):   # BASELINE configs 4 / 3 (shared cotangents as in the parity tests)
        d = P.seir_batch(B) if name == "seir" else (P.robertson_batch(B) if name == "robertson" else P.robertson5_batch(B))
        n = d["y0"].shape[1]
        if name != "seir":
            d["ps"], d["pr"] = d["params"], np.zeros(0)
        d["grads"] = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(d["tvals"]))[:, None] + 0.9 * np.arange(n)[None, :])
        return d
    return batch_of(name, B)


def main():
    from sunode_amd import _native
    from sunode_amd.solver import AdjointSolver
    from tools.problem_cache import make_problem
    name, B = sys.argv[1], int(sys.argv[2])
    prob = make_problem(name)
    d = batch(name, B)
    tol = dict(abstol=d["atol"], reltol=d["rtol"], backward_abstol=d["atol"], backward_reltol=d["rtol"],
               quad_abstol=d["atol"], quad_reltol=d["rtol"])
    sol = AdjointSolver(prob, **tol)
    tv = d["tvals"]
    for _ in range(3):
        y, st, stats = sol.solve_forward_batch(d["t0"], tv, d["y0"], d["ps"], d["pr"])
        g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], d["t0"], tv, d["grads"])
    f, b = sol._engine().last_kernel_ms()
    fam = _native.kernel_variant(prob.native_source())
    print("%s (n = %d, p = %d) B = %d, %s, %d lane(s) per instance: forward %.2f ms, backward %.2f ms -> %.4g solves/s; "
          "failed %d; steps %.0f + %.0f" % (name, prob.n_states, prob.n_params, B, fam[0], fam[1], f, b,
                                            B / ((f + b) * 1e-3), int((st != 0).sum() + (stb != 0).sum()),
                                            stats[:, 0].mean(), statsb[:, 0].mean()))


if __name__ == "__main__":
    main()
