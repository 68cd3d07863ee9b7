"""What would ordering the instances of a batch by their amount of work buy?  Runs forward + adjoint on the batch as
generated, then on the same batch permuted so that instances with similar forward step counts share a wavefront
(and, as an upper bound, sorted by the backward attempt count itself).
python tools/ab_sorted.py lv 65536 | robertson 262144 | seir 16384"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sunode_amd import SympyProblem  # noqa: E402
from sunode_amd.solver import AdjointSolver  # noqa: E402
from tools.problems import PROBLEMS, lv_batch, robertson_batch, seir_batch  # noqa: E402

name, B = sys.argv[1], int(sys.argv[2])
s = PROBLEMS[name]
prob = SympyProblem(s["params"], s["states"], s["rhs"], s["derivative_params"])
if name == "lv":
    d = lv_batch(B); ps, pr = d["params"][:, :2], d["params"][:, 2:]; rt, at = 1e-8, 1e-8
    grads = np.ones((50, 2))
elif name == "seir":
    d = seir_batch(B); ps, pr = d["ps"], d["pr"]; rt, at = 1e-8, 1e-8
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(d["tvals"]))[:, None] + 0.9 * np.arange(16)[None, :])
else:
    d = robertson_batch(B); ps, pr = d["params"], np.zeros(0); rt, at = 1e-8, 1e-10
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(7)[:, None] + 0.9 * np.arange(3)[None, :])
tv = d["tvals"]
y0 = d["y0"]
per_instance_pr = pr.ndim == 2
sol = AdjointSolver(prob, abstol=at, reltol=rt, backward_abstol=at, backward_reltol=rt, quad_abstol=at, quad_reltol=rt)


def run(perm, tag):
    a_y0, a_ps = y0[perm], ps[perm]
    a_pr = pr[perm] if per_instance_pr else pr
    for rep in range(3):
        y, st, sf = sol.solve_forward_batch(0.0, tv, a_y0, a_ps, a_pr)
        g, lam, stb, sb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
        f, b = sol._engine().last_kernel_ms()
    print("%s B=%d %-28s fwd %.2f ms, bwd %.2f ms -> %.4g solves/s" % (name, B, tag, f, b, B / ((f + b) * 1e-3)), flush=True)
    inv = np.argsort(perm)
    return sf[inv], sb[inv], g[inv]


ident = np.arange(B)
sf, sb, g0 = run(ident, "as generated")
_, _, g1 = run(np.argsort(sf[:, 8], kind="stable"), "sorted by forward points")
assert np.array_equal(g0, g1)
run(np.argsort(sf[:, 0] * 4096 + sb[:, 0] % 4096, kind="stable"), "sorted by fwd steps, bwd steps")
run(np.argsort(sb[:, 0], kind="stable"), "sorted by backward steps")
run(np.random.default_rng(0).permutation(B), "random permutation")


def wave_order(key, descending=True):
    """permutation that keeps every wavefront (64 consecutive instances) together and orders the wavefronts by key"""
    W = B // 64
    k = key[:W * 64].reshape(W, 64).max(axis=1)
    order = np.argsort(-k if descending else k, kind="stable")
    perm = (order[:, None] * 64 + np.arange(64)[None, :]).ravel()
    return np.concatenate([perm, np.arange(W * 64, B)])


# launch order of the wavefronts (longest first = LPT list scheduling over the SIMDs), instances stay in their wavefront
run(wave_order(sb[:, 14]), "waves: most bwd attempts first")
run(wave_order(sf[:, 8]), "waves: most fwd points first")
run(wave_order(sb[:, 14], False), "waves: fewest bwd attempts first")
run(ident, "as generated (again)")
