"""Generate tests/golden/truth_*.npz and tests/golden/dvode_stats.json.

Independent oracles for the integrator half:

* ``dvode_stats.json``: scipy.integrate.ode('vode', method='bdf') -- Fortran DVODE, the
  direct ancestor of CVODE -- step statistics and end states on Lotka-Volterra /
  Robertson.  The CVODE forward controller reproduces these counters exactly on LV.
* ``truth_<name>.npz``: tight-tolerance solutions (DOP853 / Radau, rtol 1e-13) of
  the ODE *augmented with its forward sensitivity equations* dS/dt = J S + df/dp
  (and dS0/dt = J S0 for the initial-value sensitivities), giving y(t_k) and the
  exact gradients of L = sum_k sum_i g[k,i] y_i(t_k) w.r.t. the differentiated
  parameters and y0 -- what ``solve_backward`` returns as ``grad_out`` and
  ``-lamda_out`` (/root/reference/sunode/solver.py:783-784,
  wrappers/as_pytensor.py:294-308).

Usage: python tools/make_golden_truth.py   (a few minutes; outputs are committed)
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import sympy as sym
from scipy.integrate import ode, solve_ivp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from sunode_amd import SympyProblem  # noqa: E402
from sunode_amd.symode.problem import HOST_FUNCTIONS  # noqa: E402
from tools.problems import (EXTRA_PROBLEMS, PROBLEMS, forcing_batch, logistic_switch_batch, lv_batch,  # noqa: E402
                            misc_batch, robertson_batch, seir_batch)

GOLD = os.path.join(ROOT, "tests", "golden")


def make(name):
    s = {**PROBLEMS, **EXTRA_PROBLEMS}[name]
    return SympyProblem(s["params"], s["states"], s["rhs"], s["derivative_params"])


def augmented_rhs(prob):
    """Callable f(t, z, ps, pr) for z = [y, S (n x p, column-major by param), S0 (n x n)]."""
    n, p = prob.n_states, prob.n_params
    y = list(prob._sym_statevec)
    ps = list(prob._sym_deriv_paramsvec)
    pr = list(prob._sym_fixed_paramsvec)
    f = sym.lambdify([prob._sym_time, y, ps, pr], list(prob._sym_dydt), modules=[HOST_FUNCTIONS, "numpy"], cse=True)
    J = sym.lambdify([prob._sym_time, y, ps, pr], sym.Matrix(prob._sym_dydt_jac), modules=[HOST_FUNCTIONS, "numpy"], cse=True)
    P = sym.lambdify([prob._sym_time, y, ps, pr], sym.Matrix(prob._sym_dydp), modules=[HOST_FUNCTIONS, "numpy"], cse=True) if p else None

    def rhs(t, z, psv, prv):
        yv = z[:n]
        S = z[n:n + n * p].reshape(p, n).T
        S0 = z[n + n * p:].reshape(n, n).T
        Jv = np.asarray(J(t, yv, psv, prv), dtype=float).reshape(n, n)
        out = np.empty_like(z)
        out[:n] = np.asarray(f(t, yv, psv, prv), dtype=float)
        if p:
            Pv = np.asarray(P(t, yv, psv, prv), dtype=float).reshape(n, p)
            out[n:n + n * p] = (Jv @ S + Pv).T.ravel()
        out[n + n * p:] = (Jv @ S0).T.ravel()
        return out
    return rhs


def truth_batch(prob, y0, ps, pr, t0, tvals, grads, method, rtol=1e-13, atol=1e-15):
    n, p = prob.n_states, prob.n_params
    B = y0.shape[0]
    rhs = augmented_rhs(prob)
    y_out = np.zeros((B, len(tvals), n))
    grad_p = np.zeros((B, p))
    grad_y0 = np.zeros((B, n))
    for b in range(B):
        prb = pr if pr.ndim == 1 else pr[b]
        z0 = np.concatenate([y0[b], np.zeros(n * p), np.eye(n).ravel()])
        sol = solve_ivp(rhs, (t0, tvals[-1]), z0, method=method, t_eval=tvals, args=(ps[b], prb),
                        rtol=rtol, atol=atol)
        assert sol.success, sol.message
        z = sol.y.T
        y_out[b] = z[:, :n]
        g = grads if grads.ndim == 2 else grads[b]
        S = z[:, n:n + n * p].reshape(len(tvals), p, n)
        S0 = z[:, n + n * p:].reshape(len(tvals), n, n)        # [k, j(y0 index), i(state)]
        grad_p[b] = np.einsum("ki,kpi->p", g, S)
        grad_y0[b] = np.einsum("ki,kji->j", g, S0)
        print("  truth instance", b, "nfev", sol.nfev, flush=True)
    return y_out, grad_p, grad_y0


def cotangent(n_t, n):
    """Non-degenerate dL/dy_out: Robertson and SEIR conserve sum(y), so grads = ones
    (BASELINE's loss = sum(y_out)) has an identically-zero parameter gradient there."""
    k = np.arange(n_t)[:, None]
    i = np.arange(n)[None, :]
    return 1.0 + 0.5 * np.cos(1.7 * k + 0.9 * i)


def dvode_run(f, jac, y0, tvals, rtol, atol, args):
    r = ode(f, jac).set_integrator("vode", method="bdf", with_jacobian=True, rtol=rtol, atol=atol, nsteps=100000)
    r.set_initial_value(y0, tvals[0]).set_f_params(*args).set_jac_params(*args)
    ys = [np.array(y0, float)]
    for t in tvals[1:]:
        ys.append(r.integrate(t).copy())
        assert r.successful()
    iw = r._integrator.iwork
    return dict(nst=int(iw[10]), nfe=int(iw[11]), nje=int(iw[12]), qlast=int(iw[13]), nlu=int(iw[18]),
                nni=int(iw[19]), ncfn=int(iw[20]), netf=int(iw[21]), y=np.array(ys).tolist())


def transcendental_truth():
    """truth_<name>.npz for the models with transcendental right-hand sides (8 draws each, per-instance cotangents):
    ``forcing`` (expit / logaddexp / spline input -- the B-spline evaluated by the Cox-de Boor recursion, i.e. NOT by
    the polynomial pieces the generated code shares with the reference), ``logistic_switch``, ``misc``."""
    for name, batch in (("forcing", forcing_batch), ("logistic_switch", logistic_switch_batch), ("misc", misc_batch)):
        prob = make(name)
        d = batch(8)
        y_out, gp, gy0 = truth_batch(prob, d["y0"], d["ps"], d["pr"], d["t0"], d["tvals"], d["grads"], "DOP853")
        np.savez(os.path.join(GOLD, "truth_%s.npz" % name), y0=d["y0"], ps=d["ps"], pr=d["pr"], t0=d["t0"],
                 tvals=d["tvals"], grads=d["grads"], y_out=y_out, grad_params=gp, grad_y0=gy0)


def sweep_truth():
    """truth_sweep_<name>.npz for two shapes of the parity sweep (tests/test_shape_sweep.py): ``lv12`` (2 states, 12
    differentiated parameters) and ``rn12_4`` (12 states): 4 draws each of the sweep's own batch."""
    from tools.problem_cache import spec_of
    from tools.sweep_cases import batch_of
    for name in ("lv12", "rn12_4"):
        s = spec_of(name)
        prob = SympyProblem(s["params"], s["states"], s["rhs"], s["derivative_params"])
        d = batch_of(name, 4)
        y_out, gp, gy0 = truth_batch(prob, d["y0"], d["ps"], d["pr"], d["t0"], d["tvals"], d["grads"], "DOP853")
        np.savez(os.path.join(GOLD, "truth_sweep_%s.npz" % name), y0=d["y0"], ps=d["ps"], pr=d["pr"], t0=d["t0"],
                 tvals=d["tvals"], grads=d["grads"], y_out=y_out, grad_params=gp, grad_y0=gy0)


def main():
    os.makedirs(GOLD, exist_ok=True)
    if "--transcendental" in sys.argv:          # only the round-6 fixtures (the others are unchanged)
        transcendental_truth()
        return
    if "--sweep" in sys.argv:
        sweep_truth()
        return
    # ---------------- DVODE statistics ----------------
    def lv_f(t, y, a, b, c, d):
        return [a * y[0] - b * y[0] * y[1], d * y[0] * y[1] - c * y[1]]

    def lv_j(t, y, a, b, c, d):
        return [[a - b * y[1], -b * y[0]], [d * y[1], d * y[0] - c]]

    def rob_f(t, y, k1, k2, k3):
        return [-k1 * y[0] + k2 * y[1] * y[2], k1 * y[0] - k2 * y[1] * y[2] - k3 * y[1] ** 2, k3 * y[1] ** 2]

    def rob_j(t, y, k1, k2, k3):
        return [[-k1, k2 * y[2], k2 * y[1]], [k1, -k2 * y[2] - 2 * k3 * y[1], -k2 * y[1]], [0.0, 2 * k3 * y[1], 0.0]]

    stats = {}
    tv = np.linspace(0, 10)
    for tol in (1e-8, 1e-10):
        stats["lv_readme_%g" % tol] = dict(rtol=tol, atol=tol, tvals=tv.tolist(), y0=[1.0, 0.1],
                                           params=[0.1, 0.2, 0.3, 0.4],
                                           **dvode_run(lv_f, lv_j, [1.0, 0.1], tv, tol, tol, (0.1, 0.2, 0.3, 0.4)))
    lvb = lv_batch(8)
    for b in range(8):
        stats["lv_batch_%d" % b] = dict(rtol=1e-8, atol=1e-8, tvals=lvb["tvals"].tolist(), y0=lvb["y0"][b].tolist(),
                                        params=lvb["params"][b].tolist(),
                                        **dvode_run(lv_f, lv_j, lvb["y0"][b], lvb["tvals"], 1e-8, 1e-8, tuple(lvb["params"][b])))
    rb = robertson_batch(4)
    tv_long = np.array([0.0] + [0.4 * 10.0 ** k for k in range(12)])
    stats["robertson_4e10"] = dict(rtol=1e-8, atol=1e-10, tvals=tv_long.tolist(), y0=[1.0, 0.0, 0.0],
                                   params=[0.04, 1e4, 3e7],
                                   **dvode_run(rob_f, rob_j, [1.0, 0.0, 0.0], tv_long, 1e-8, 1e-10, (0.04, 1e4, 3e7)))
    stats["robertson_4e4"] = dict(rtol=1e-8, atol=1e-10, tvals=rb["tvals"].tolist(), y0=[1.0, 0.0, 0.0],
                                  params=[0.04, 1e4, 3e7],
                                  **dvode_run(rob_f, rob_j, [1.0, 0.0, 0.0], rb["tvals"], 1e-8, 1e-10, (0.04, 1e4, 3e7)))
    # step-by-step DVODE trace (ITASK=2) of the stiff transient: times and orders of the first steps
    import warnings
    r = ode(rob_f, rob_j).set_integrator("vode", method="bdf", with_jacobian=True, rtol=1e-8, atol=1e-10,
                                         nsteps=100000)
    r.set_initial_value([1.0, 0.0, 0.0], 0.0).set_f_params(0.04, 1e4, 3e7).set_jac_params(0.04, 1e4, 3e7)
    trace_t, trace_q = [], []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        while r.t < 40.0:
            r.integrate(40.0, step=True)
            trace_t.append(float(r.t))
            trace_q.append(int(r._integrator.iwork[13]))
    iw = r._integrator.iwork
    stats["robertson_trace_T40"] = dict(t=trace_t, q=trace_q, nst=int(iw[10]), nfe=int(iw[11]), nje=int(iw[12]),
                                        nlu=int(iw[18]), nni=int(iw[19]), ncfn=int(iw[20]), netf=int(iw[21]))
    with open(os.path.join(GOLD, "dvode_stats.json"), "w") as fh:
        json.dump(stats, fh)
    print("dvode:", {k: (v["nst"], v["nfe"], v["nje"], v["nlu"], v["nni"], v["ncfn"], v["netf"])
                     for k, v in stats.items()})

    # ---------------- truth: LV config-2 batch (16 instances) ----------------
    prob = make("lv")
    B = 16
    d = lv_batch(B)
    ps = d["params"][:, prob.params_subset.subset_index]
    pr = d["params"][:, prob.params_subset.remainder_index]
    g = np.ones((len(d["tvals"]), 2))
    y_out, gp, gy0 = truth_batch(prob, d["y0"], ps, pr, d["t0"], d["tvals"], g, "DOP853")
    np.savez(os.path.join(GOLD, "truth_lv.npz"), y0=d["y0"], ps=ps, pr=pr, t0=d["t0"], tvals=d["tvals"],
             grads=g, y_out=y_out, grad_params=gp, grad_y0=gy0)

    # README instance with non-trivial cotangents
    rng_g = np.cos(np.arange(100.0)).reshape(50, 2)
    y_out, gp, gy0 = truth_batch(prob, np.array([[1.0, 0.1]]), np.array([[0.1, 0.2]]), np.array([[0.3, 0.4]]),
                                 0.0, np.linspace(0, 10), rng_g, "DOP853")
    np.savez(os.path.join(GOLD, "truth_lv_readme.npz"), y0=np.array([[1.0, 0.1]]), ps=np.array([[0.1, 0.2]]),
             pr=np.array([[0.3, 0.4]]), t0=0.0, tvals=np.linspace(0, 10), grads=rng_g, y_out=y_out,
             grad_params=gp, grad_y0=gy0)

    # ---------------- truth: Robertson config-3 (4 instances, Radau) ----------------
    prob = make("robertson")
    d = robertson_batch(4)
    g = cotangent(len(d["tvals"]), 3)
    y_out, gp, gy0 = truth_batch(prob, d["y0"], d["params"], np.zeros((4, 0)), d["t0"], d["tvals"], g, "Radau",
                                 rtol=1e-12, atol=1e-16)
    np.savez(os.path.join(GOLD, "truth_robertson.npz"), y0=d["y0"], ps=d["params"], pr=np.zeros((4, 0)),
             t0=d["t0"], tvals=d["tvals"], grads=g, y_out=y_out, grad_params=gp, grad_y0=gy0)

    # ---------------- truth: SEIR config-4 (2 instances) ----------------
    prob = make("seir")
    d = seir_batch(2)
    g = cotangent(len(d["tvals"]), 16)
    y_out, gp, gy0 = truth_batch(prob, d["y0"], d["ps"], d["pr"], d["t0"], d["tvals"], g, "DOP853")
    np.savez(os.path.join(GOLD, "truth_seir.npz"), y0=d["y0"], ps=d["ps"], pr=d["pr"], t0=d["t0"],
             tvals=d["tvals"], grads=g, y_out=y_out, grad_params=gp, grad_y0=gy0)
    transcendental_truth()
    sweep_truth()
    print("done")


if __name__ == "__main__":
    main()
