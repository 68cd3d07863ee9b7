import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from tests.helpers import make_problem
from sunode_amd.solver import AdjointSolver
from tools.problems import lv_batch
prob = make_problem("lv"); B = 65536
d = lv_batch(B); ps, pr = d["params"][:, :2], d["params"][:, 2:]
tv = d["tvals"]; grads = np.ones((len(tv), 2))
for interp in ("polynomial", "hermite"):
    sol = AdjointSolver(prob, abstol=1e-8, reltol=1e-8, backward_abstol=1e-8, backward_reltol=1e-8, quad_abstol=1e-8, quad_reltol=1e-8, interpolation=interp)
    for rep in range(3):
        y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
        g, lam, stb, sb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    f, b = sol._engine().last_kernel_ms()
    print(interp, "fwd %.2f bwd %.2f ms -> %.3g solves/s" % (f, b, B / ((f + b) * 1e-3)), int((st != 0).sum() + (stb != 0).sum()))
