import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from sunode_amd import SympyProblem, _native
from sunode_amd.solver import AdjointSolver
from tools.problems import network100, network_batch
s = network100()
prob = SympyProblem(s["params"], s["states"], s["rhs"], s["derivative_params"])
B = 1024
d = network_batch(B)
tv = d["tvals"]
grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(prob.n_states)[None, :])
real = _native.code_object_path
for tag in sys.argv[1:]:
    _native.code_object_path = lambda *a, **k: os.path.join(os.getcwd(), "sunode_amd/_cache/sa_%s.hsaco" % tag)
    sol = AdjointSolver(prob, abstol=1e-8, reltol=1e-8, backward_abstol=1e-8, backward_reltol=1e-8, quad_abstol=1e-8, quad_reltol=1e-8, max_steps=1024)
    res = []
    for rep in range(4):
        y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], d["ps"], d["pr"])
        g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
        f, b = sol._engine().last_kernel_ms()
        res.append((round(f, 1), round(b, 1)))
    print(tag, res, "%.0f solves/s" % (B / ((res[-1][0] + res[-1][1]) * 1e-3)), float(np.abs(g).sum()))
