import sys, numpy as np
sys.path.insert(0, '.')
from tests.helpers import make_problem
from sunode_amd.solver import AdjointSolver
from tools.problems import robertson_batch
prob = make_problem("robertson"); d = robertson_batch(512); tv = d["tvals"]
for compact in (False, True):
    sol = AdjointSolver(prob, abstol=1e-10, reltol=1e-8, backward_abstol=1e-8, backward_reltol=1e-8, quad_abstol=1e-8, quad_reltol=1e-8, compact_trajectory=compact)
    y, st, sf = sol.solve_forward_batch(0.0, tv, d["y0"], d["params"], np.zeros(0))
    g, lam, stb, sb = sol.solve_backward_batch(tv[-1], 0.0, tv, np.ones((len(tv), 3)))
    print("compact", compact, "fwd fail", int((st != 0).sum()), "bwd status", np.unique(stb, return_counts=True), "bwd steps", sb[:, 0].mean(), "sum g", float(np.nansum(np.abs(g))))
