"""Third-party pins at n = 100 / n = 24 and for mid-size forward sensitivities (VERDICT r3 "do this" #2).

    python tools/make_golden_network.py          (a few minutes; outputs are committed)

Everything here is a plain numpy restatement of the models in tools/problems.py -- no generated code, no sympy, no
import of the engine or of the oracle -- integrated by codes that are not ours:

* ``tests/golden/dvode_network.json``: Fortran DVODE (scipy ``ode('vode', method='bdf')``, analytic Jacobian) on four
  draws each of the 100-state network (BASELINE config 5) and of the 24-state one: every counter (steps, rhs
  evaluations, Jacobians, LU set-ups, Newton iterations, convergence / error-test failures, last order) and the
  states at the output times.  Pins the dense LU path at n = 100 (reference: ``SUNLinSol_Dense``,
  /root/reference/sunode/solver.py:601,618) and the structured ``SA_MATVEC`` callbacks through the controller.
* ``tests/golden/truth_network100.npz`` / ``truth_network24.npz``: DOP853 (rtol 1e-12) on the ODE augmented with its
  sensitivity equations dS/dt = J S + df/dp, dS0/dt = J S0: states and the exact gradients of
  L = sum_k g_k . y(t_k) with a non-trivial cotangent -- what solve_backward returns as grad_out / -lamda_out
  (/root/reference/sunode/solver.py:783-784).
* ``tests/golden/truth_sens_seir.npz``: SEIR (n = 16, 8 differentiated parameters) forward-sensitivity truth
  dy(t_k)/dp (16 x 8 per output time), the reference's ``Solver(sens_mode=...)`` output
  (/root/reference/sunode/solver.py:360-392, 467-527).
"""
import json
import os
import sys

import numpy as np
from scipy.integrate import ode, solve_ivp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.problems import network_batch, seir_batch  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


# ---- network (tools/problems.py::make_network): x_i' = sum_j K_ij x_j - s0 x_i sum_j K_ji - s1 x_i T/(s2 + T) + s3
def net_f(t, x, K, s):
    T = x.sum()
    return K @ x - s[0] * x * K.sum(axis=0) - s[1] * x * T / (s[2] + T) + s[3]


def net_j(t, x, K, s):
    T = x.sum()
    J = K - np.outer(s[1] * x * s[2] / (s[2] + T) ** 2, np.ones(len(x)))
    J[np.diag_indices(len(x))] -= s[0] * K.sum(axis=0) + s[1] * T / (s[2] + T)
    return J


def net_dfdp(t, x, K, s):
    T = x.sum()
    return np.stack([-x * K.sum(axis=0), -x * T / (s[2] + T), s[1] * x * T / (s[2] + T) ** 2, np.ones(len(x))], axis=1)


# ---- SEIR (tools/problems.py::seir); differentiated: beta[0..3], sigma, gamma, mu, nu (mu, nu unused)
def seir_f(t, y, beta, C, sigma, gamma):
    S, E, I, R = y[0:4], y[4:8], y[8:12], y[12:16]
    lam = beta * (C @ (I / (S + E + I + R)))
    return np.concatenate([-lam * S, lam * S - sigma * E, sigma * E - gamma * I, gamma * I])


def seir_j(t, y, beta, C, sigma, gamma):
    S, E, I, R = y[0:4], y[4:8], y[8:12], y[12:16]
    N = S + E + I + R
    lam = beta * (C @ (I / N))
    dlam = np.zeros((4, 16))
    for j in range(4):
        c = beta * C[:, j]
        for blk, v in ((0, -I[j] / N[j] ** 2), (4, -I[j] / N[j] ** 2), (8, 1.0 / N[j] - I[j] / N[j] ** 2),
                       (12, -I[j] / N[j] ** 2)):
            dlam[:, blk + j] += c * v
    J = np.zeros((16, 16))
    for i in range(4):
        J[i] += -S[i] * dlam[i]; J[i, i] += -lam[i]
        J[4 + i] += S[i] * dlam[i]; J[4 + i, i] += lam[i]; J[4 + i, 4 + i] += -sigma
        J[8 + i, 4 + i] += sigma; J[8 + i, 8 + i] += -gamma
        J[12 + i, 8 + i] += gamma
    return J


def seir_dfdp(t, y, beta, C, sigma, gamma):
    S, E, I, R = y[0:4], y[4:8], y[8:12], y[12:16]
    x = C @ (I / (S + E + I + R))
    P = np.zeros((16, 8))
    for i in range(4):
        P[i, i] = -x[i] * S[i]; P[4 + i, i] = x[i] * S[i]
        P[4 + i, 4] = -E[i]; P[8 + i, 4] = E[i]
        P[8 + i, 5] = -I[i]; P[12 + i, 5] = I[i]
    return P


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


def augmented(f, jac, dfdp, n, p):
    def rhs(t, z, *args):
        y = z[:n]
        S = z[n:n + n * p].reshape(p, n).T
        S0 = z[n + n * p:].reshape(n, n).T
        J = jac(t, y, *args)
        out = np.empty_like(z)
        out[:n] = f(t, y, *args)
        out[n:n + n * p] = (J @ S + dfdp(t, y, *args)).T.ravel()
        out[n + n * p:] = (J @ S0).T.ravel()
        return out
    return rhs


def cotangent(n_t, n):
    return 1.0 + 0.5 * np.cos(1.7 * np.arange(n_t)[:, None] + 0.9 * np.arange(n)[None, :])


def truth(f, jac, dfdp, n, p, y0, tvals, args, rtol, atol):
    z0 = np.concatenate([y0, np.zeros(n * p), np.eye(n).ravel()])
    sol = solve_ivp(augmented(f, jac, dfdp, n, p), (tvals[0], tvals[-1]), z0, method="DOP853", t_eval=tvals,
                    args=args, rtol=rtol, atol=atol)
    assert sol.success, sol.message
    z = sol.y.T
    return z[:, :n], z[:, n:n + n * p].reshape(len(tvals), p, n), z[:, n + n * p:].reshape(len(tvals), n, n), sol.nfev


def main():
    os.makedirs(GOLD, exist_ok=True)
    stats = {}
    for n in (100, 24):
        d = network_batch(4, n=n)
        K = d["pr"].reshape(n, n)
        tv = d["tvals"] if n == 100 else d["tvals"][:6]
        for b in range(4):
            stats["network%d_batch_%d" % (n, b)] = dict(
                n=n, rtol=1e-8, atol=1e-8, tvals=tv.tolist(), ps=d["ps"][b].tolist(),
                **dvode_run(net_f, net_j, d["y0"][b], tv, 1e-8, 1e-8, (K, d["ps"][b])))
            c = stats["network%d_batch_%d" % (n, b)]
            print("dvode network%d draw %d:" % (n, b), tuple(c[k] for k in ("nst", "nfe", "nje", "nlu", "nni", "ncfn", "netf", "qlast")), flush=True)
    with open(os.path.join(GOLD, "dvode_network.json"), "w") as fh:
        json.dump(stats, fh)

    for n, B in ((100, 2), (24, 3)):
        d = network_batch(B, n=n)
        K = d["pr"].reshape(n, n)
        tv = d["tvals"] if n == 100 else d["tvals"][:6]
        g = cotangent(len(tv), n)
        y_out = np.zeros((B, len(tv), n)); gp = np.zeros((B, 4)); gy0 = np.zeros((B, n))
        for b in range(B):
            y, S, S0, nfev = truth(net_f, net_j, net_dfdp, n, 4, d["y0"][b], tv, (K, d["ps"][b]), 1e-12, 1e-14)
            y_out[b] = y
            gp[b] = np.einsum("ki,kpi->p", g, S)
            gy0[b] = np.einsum("ki,kji->j", g, S0)
            print("truth network%d draw %d nfev %d" % (n, b, nfev), flush=True)
        np.savez(os.path.join(GOLD, "truth_network%d.npz" % n), y0=d["y0"], ps=d["ps"], t0=0.0, tvals=tv, grads=g,
                 y_out=y_out, grad_params=gp, grad_y0=gy0)       # (the rate matrix is network_batch's: not stored)

    d = seir_batch(2)
    C = d["pr"].reshape(4, 4)
    tv = d["tvals"][::5]
    y_out = np.zeros((2, len(tv), 16)); sens = np.zeros((2, len(tv), 8, 16))
    for b in range(2):
        ps = d["ps"][b]
        y, S, _, nfev = truth(seir_f, seir_j, seir_dfdp, 16, 8, d["y0"][b], tv, (ps[:4], C, ps[4], ps[5]), 1e-13, 1e-13)
        y_out[b], sens[b] = y, S
        print("truth seir sens draw %d nfev %d" % (b, nfev), flush=True)
    np.savez(os.path.join(GOLD, "truth_sens_seir.npz"), y0=d["y0"], ps=d["ps"], pr=d["pr"], t0=0.0, tvals=tv,
             y_out=y_out, sens=sens)


if __name__ == "__main__":
    main()
