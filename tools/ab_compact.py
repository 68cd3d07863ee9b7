"""A/B of the arena record formats of the one-lane-per-instance kernel: table records (8 + 6n doubles per stored step,
built by the forward kernel) against compact records ({order, t, y[n]}, table rebuilt by the backward kernel).
python tools/ab_compact.py robertson 262144 | lv 65536 | seir 16384"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sunode_amd import SympyProblem  # noqa: E402
from sunode_amd.solver import AdjointSolver  # noqa: E402
from tools.problems import EXTRA_PROBLEMS, PROBLEMS, lv_batch, network100, network_batch, robertson_batch, seir_batch  # noqa: E402

name, B = sys.argv[1], int(sys.argv[2])
only = sys.argv[3] if len(sys.argv) > 3 else None
s = network100() if name == "network100" else {**PROBLEMS, **EXTRA_PROBLEMS}[name]
prob = SympyProblem(s["params"], s["states"], s["rhs"], s["derivative_params"])
if name == "lv":
    d = lv_batch(B); ps, pr = d["params"][:, :2], d["params"][:, 2:]; rt, at = 1e-8, 1e-8
    grads = np.ones((50, 2))
elif name.startswith("network"):
    nn = int(name[7:])
    d = network_batch(B, nn); ps, pr = d["ps"], d["pr"]; rt, at = d["rtol"], d["atol"]
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(d["tvals"]))[:, None] + 0.9 * np.arange(nn)[None, :])
elif name == "seir":
    d = seir_batch(B); ps, pr = d["ps"], d["pr"]; rt, at = 1e-8, 1e-8
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(d["tvals"]))[:, None] + 0.9 * np.arange(16)[None, :])
else:
    d = robertson_batch(B); ps, pr = d["params"], np.zeros(0); rt, at = 1e-8, 1e-10
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(7)[:, None] + 0.9 * np.arange(3)[None, :])
tv = d["tvals"]
for compact in (False, True):
    if only and only != ("compact" if compact else "table"):
        continue
    sol = AdjointSolver(prob, abstol=at, reltol=rt, backward_abstol=at, backward_reltol=rt, quad_abstol=at,
                        quad_reltol=rt, compact_trajectory=compact)
    for rep in range(3):
        y, st, sf = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
        g, lam, stb, sb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
        f, b = sol._engine().last_kernel_ms()
    info = sol._engine().arena_info()
    print("%s B=%d %s records: fwd %.2f ms, bwd %.2f ms -> %.3g solves/s; arena %.2f GB%s; points %.0f, rebuilds %.0f, sum(g) %.17g"
          % (name, B, "compact" if compact else "table  ", f, b, B / ((f + b) * 1e-3), info[0] / 1e9,
             " (tiled)" if info[2] else "", sf[:, 8].mean(), sb[:, 12].mean(), float(np.abs(g).sum())))
    sol._engine().close()
