"""Generate tests/golden/callbacks_network.json FROM THE REFERENCE: the dense reaction networks (24 and 100 states).

Same recipe as tools/make_golden_callbacks.py (the reference's symbolic half -- dtypesubset.py, symode/lambdify.py,
symode/problem.py -- imported under stub modules, its generated Python callbacks evaluated at fixed points), for the
two problems whose callbacks this build emits in STRUCTURED form (SA_MATVEC / SA_MATFILL / SA_SUM / SA_ROLLED,
symode/codegen.py): until now they were only checked against a sympy re-evaluation of our own expressions.

Vector callbacks (rhs, adjoint rhs, quadrature rhs) are stored in full.  The n x n matrix callbacks would be megabytes
at n = 100, so the fixture holds what pins every entry without storing it: the products M u and M^T w with two fixed
dense vectors (any wrong entry moves both), the diagonal, and a strided sample of entries.  Data only.

    python tools/make_golden_callbacks_network.py        (build container only: needs /root/reference)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.make_golden_callbacks import load_reference  # noqa: E402
from tools.problems import EXTRA_PROBLEMS, SEED, network100, std_normal  # noqa: E402


def probe_vectors(n):
    u = 1.0 + 0.5 * np.cos(0.7 * np.arange(n))
    w = 1.0 + 0.5 * np.sin(1.3 * np.arange(n) + 0.2)
    return u, w


def matrix_summary(M):
    n = M.shape[0]
    u, w = probe_vectors(n)
    idx = np.arange(0, n * n, 37)
    return dict(Mu=(M @ u).tolist(), MTw=(M.T @ w).tolist(), diag=np.diag(M).tolist(),
                sample=M.ravel()[idx].tolist())


def main():
    SympyProblem = load_reference()
    out = {}
    for name, spec, n_pts in (("network24", EXTRA_PROBLEMS["network24"], 4), ("network100", network100(), 2)):
        prob = SympyProblem(spec["params"], spec["states"], spec["rhs"], spec["derivative_params"])
        n, p = prob.n_states, prob.n_params
        n_items = prob.params_subset.n_items
        stream = 2000 + n
        y = np.exp(0.3 * std_normal(SEED, stream, n_pts * n)).reshape(n_pts, n)
        lam = std_normal(SEED, stream + 1, n_pts * n).reshape(n_pts, n)
        par = np.abs(std_normal(SEED, stream + 2, n_pts * n_items)).reshape(n_pts, n_items)
        par[:, :n * n] /= n                                   # rate matrix K ~ |N(0,1)| / n, then the 4 scales
        par[:, n * n:] = np.array([1.0, 0.5, 10.0, 0.1]) * np.exp(0.1 * std_normal(SEED, stream + 3, n_pts * 4).reshape(n_pts, 4))
        t = np.abs(std_normal(SEED, stream + 4, n_pts))
        rhs, jac = prob.make_rhs(), prob.make_jac_dense()
        adj, quad, adjjac = prob.make_adjoint_rhs(), prob.make_adjoint_quad_rhs(), prob.make_adjoint_jac_dense()
        pts = []
        for k in range(n_pts):
            ud = prob.make_user_data()
            pv = np.zeros((), dtype=prob.params_dtype)
            pv.reshape(1).view(np.float64)[:] = par[k]
            prob.update_params(ud, pv)
            yk = np.zeros((), dtype=prob.state_dtype)
            yk.reshape(1).view(np.float64)[:] = y[k]
            yk = yk.view(np.recarray)
            o_rhs = np.zeros(n); o_jac = np.zeros((n, n)); o_adj = np.zeros(n); o_quad = np.zeros(p); o_aj = np.zeros((n, n))
            codes = [int(rhs(o_rhs, t[k], yk, ud)), int(jac(o_jac, t[k], yk, None, ud)),
                     int(adj(o_adj, t[k], yk, lam[k], ud)), int(quad(o_quad, t[k], yk, lam[k], ud)),
                     int(adjjac(o_aj, t[k], yk, lam[k], None, ud))]
            pts.append(dict(t=float(t[k]), y=y[k].tolist(), lam=lam[k].tolist(), scale=par[k, n * n:].tolist(),
                            K_seed=[SEED, stream + 2, k], rhs=o_rhs.tolist(), adj=o_adj.tolist(), quad=o_quad.tolist(),
                            jac=matrix_summary(o_jac), adjjac=matrix_summary(o_aj), codes=codes))
        out[name] = dict(n=n, points=pts,
                         note="K of point k = |std_normal(SEED, stream, n_pts*n_items)|[k, :n*n] / n with (SEED, stream, k) "
                              "= K_seed (tools/problems.py::std_normal); params order: K (row-major), scale")
        print(name, "done")
    with open(os.path.join(ROOT, "tests", "golden", "callbacks_network.json"), "w") as fh:
        json.dump(out, fh)


if __name__ == "__main__":
    main()
