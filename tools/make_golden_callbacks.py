"""Generate tests/golden/callbacks.json and tests/golden/layout.json FROM THE REFERENCE.

Runs only in the build container (needs /root/reference).  It imports the
reference's *symbolic half* (dtypesubset.py, problem.py param helpers,
symode/lambdify.py, symode/problem.py) under four stub modules (numba.njit ->
identity, xarray, sunode.basic, sunode.matrix -- recipe: SURVEY.md Appendix E),
evaluates the reference-generated Python callbacks at fixed points and records:

* callbacks.json: for each problem, 8 points (t, y, lam, params) -> rhs, jac,
  adjoint rhs, quadrature rhs, adjoint jac values and return codes;
* layout.json: flat layouts the reference derives (n_states, n_params, paths,
  subset paths, flat slices, dtype itemsizes/offsets, user_data size).

The committed JSON files are data (inputs + expected outputs), no reference
source.  Usage:  python tools/make_golden_callbacks.py
"""
from __future__ import annotations

import importlib.abc  # noqa: F401  (must be imported before lambdify.py loads)
import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from tools.problems import PROBLEMS, SEED, std_normal  # noqa: E402

R = "/root/reference/sunode/"


def _stub(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    def njit(*a, **k):
        return a[0] if a and callable(a[0]) else (lambda f: f)
    _stub("numba", njit=njit)
    _stub("xarray", DataArray=type("DataArray", (), {}), Dataset=type("Dataset", (), {}))
    pkg = _stub("sunode"); pkg.__path__ = []
    sub = _stub("sunode.symode"); sub.__path__ = []
    _stub("sunode.basic", data_dtype=np.dtype("float64"), index_dtype=np.dtype("int64"), lib=None, ffi=None)
    _stub("sunode.matrix", Sparse=object)
    pkg.dtypesubset = _load("sunode.dtypesubset", R + "dtypesubset.py")
    pkg.basic = sys.modules["sunode.basic"]
    pkg.problem = _load("sunode.problem", R + "problem.py")
    _load("sunode.symode.lambdify", R + "symode/lambdify.py")
    return _load("sunode.symode.problem", R + "symode/problem.py").SympyProblem


def positive_points(name, n_states, n_items, n_pts):
    """Deterministic evaluation points: states/params positive O(1), lam signed."""
    stream = 1000 + sum(ord(c) for c in name)
    y = np.exp(0.5 * std_normal(SEED, stream, n_pts * n_states)).reshape(n_pts, n_states)
    lam = std_normal(SEED, stream + 1, n_pts * n_states).reshape(n_pts, n_states)
    par = np.exp(0.5 * std_normal(SEED, stream + 2, n_pts * n_items)).reshape(n_pts, n_items)
    t = np.abs(std_normal(SEED, stream + 3, n_pts))
    return t, y, lam, par


def main():
    SympyProblem = load_reference()
    n_pts = 8
    out_cb, out_layout = {}, {}
    for name, spec in PROBLEMS.items():
        prob = SympyProblem(spec["params"], spec["states"], spec["rhs"], spec["derivative_params"])
        n, p = prob.n_states, prob.n_params
        n_items = prob.params_subset.n_items
        t, y, lam, par = positive_points(name, n, n_items, n_pts)
        rhs, jac = prob.make_rhs(), prob.make_jac_dense()
        adj, quad, adjjac = prob.make_adjoint_rhs(), prob.make_adjoint_quad_rhs(), prob.make_adjoint_jac_dense()
        pts = []
        for k in range(n_pts):
            ud = prob.make_user_data()
            pv = np.zeros((), dtype=prob.params_dtype)
            pv_flat = pv.reshape(1).view(np.float64)
            pv_flat[:] = par[k]
            prob.update_params(ud, pv)
            yk = np.zeros((), dtype=prob.state_dtype)
            yk.reshape(1).view(np.float64)[:] = y[k]
            yk = yk.view(np.recarray)
            o_rhs = np.zeros(n); o_jac = np.zeros((n, n)); o_adj = np.zeros(n)
            o_quad = np.zeros(p); o_adjjac = np.zeros((n, n))
            with np.errstate(all="ignore"):
                codes = [
                    int(rhs(o_rhs, t[k], yk, ud)),
                    int(jac(o_jac, t[k], yk, None, ud)),
                    int(adj(o_adj, t[k], yk, lam[k], ud)),
                    int(quad(o_quad, t[k], yk, lam[k], ud)) if p else 0,
                    int(adjjac(o_adjjac, t[k], yk, lam[k], None, ud)),
                ]
            pts.append(dict(t=float(t[k]), y=y[k].tolist(), lam=lam[k].tolist(), params=par[k].tolist(),
                            rhs=o_rhs.tolist(), jac=o_jac.tolist(), adj=o_adj.tolist(),
                            quad=o_quad.tolist(), adjjac=o_adjjac.tolist(), codes=codes))
        out_cb[name] = pts

        ps = prob.params_subset
        view = ps.subset_view_dtype
        rview = ps.remainder.subset_view_dtype
        out_layout[name] = dict(
            n_states=n, n_params=p, n_items=n_items,
            state_paths=[list(q) for q in prob.state_subset.paths],
            param_paths=[list(q) for q in ps.paths],
            subset_paths=[list(q) for q in ps.subset_paths],
            remainder_subset_paths=[list(q) for q in ps.remainder.subset_paths],
            param_slices={".".join(q): [s.start, s.stop] for q, s in ps.flat_slices.items()},
            param_shapes={".".join(q): list(s) for q, s in ps.flat_shapes.items()},
            state_slices={".".join(q): [s.start, s.stop] for q, s in prob.state_subset.flat_slices.items()},
            params_itemsize=prob.params_dtype.itemsize,
            state_itemsize=prob.state_dtype.itemsize,
            subset_itemsize=ps.subset_dtype.itemsize,
            subset_view_offsets=[int(view.fields[nm][1]) for nm in (view.names or ())],
            remainder_view_offsets=[int(rview.fields[nm][1]) for nm in (rview.names or ())],
            user_data_itemsize=prob.user_data_dtype.itemsize,
        )
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    with open(os.path.join(gold, "callbacks.json"), "w") as fh:
        json.dump(out_cb, fh)
    with open(os.path.join(gold, "layout.json"), "w") as fh:
        json.dump(out_layout, fh, indent=1)
    print("wrote", {k: len(v) for k, v in out_cb.items()})


if __name__ == "__main__":
    main()
