"""Generate tests/golden/dvode_seir.json: Fortran DVODE (scipy.integrate.ode 'vode', method='bdf', analytic
Jacobian) on the SEIR problem of BASELINE config 4 (n = 16) -- an independent counter / step-trace pin for the
mid-size mappings (cooperative and lane-group kernels), as dvode_stats.json is for LV and Robertson.

    python tools/make_golden_dvode_seir.py

The right-hand side and Jacobian are plain numpy restatements of the model in tools/problems.py (not generated
code), so the fixture does not depend on the code generator it helps to pin.
"""
import json
import os
import sys
import warnings

import numpy as np
from scipy.integrate import ode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.problems import seir_batch  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def seir_f(t, y, beta, C, sigma, gamma):
    """tools/problems.py::seir: force_i = sum_j beta_i C_ij I_j / N_j; S' = -force S, E' = force S - sigma E,
    I' = sigma E - gamma I, R' = gamma I (the rates mu, nu are parameters of the model record but unused)."""
    S, E, I, R = y[0:4], y[4:8], y[8:12], y[12:16]
    N = S + E + I + R
    lam = beta * (C @ (I / N))
    return np.concatenate([-lam * S, lam * S - sigma * E, sigma * E - gamma * I, gamma * I])


def seir_j(t, y, beta, C, sigma, gamma):
    n = 16
    J = np.zeros((n, n))
    S, E, I, R = y[0:4], y[4:8], y[8:12], y[12:16]
    N = S + E + I + R
    lam = beta * (C @ (I / N))
    # x_j = I_j / N_j: dx_j/dI_j = 1/N_j - I_j/N_j^2, dx_j/d{S,E,R}_j = -I_j/N_j^2
    dlam = np.zeros((4, n))
    for i in range(4):
        for j in range(4):
            c = beta[i] * C[i, j]
            dI = 1.0 / N[j] - I[j] / N[j] ** 2
            dO = -I[j] / N[j] ** 2
            dlam[i, 0 + j] += c * dO
            dlam[i, 4 + j] += c * dO
            dlam[i, 8 + j] += c * dI
            dlam[i, 12 + j] += c * dO
    for i in range(4):
        J[i, :] += -S[i] * dlam[i]
        J[i, i] += -lam[i]
        J[4 + i, :] += S[i] * dlam[i]
        J[4 + i, i] += lam[i]
        J[4 + i, 4 + i] += -sigma
        J[8 + i, 4 + i] += sigma
        J[8 + i, 8 + i] += -gamma
        J[12 + i, 8 + i] += gamma
    return J


def run(y0, tvals, rtol, atol, args, trace_until=None):
    r = ode(seir_f, seir_j).set_integrator("vode", method="bdf", with_jacobian=True, rtol=rtol, atol=atol,
                                           nsteps=100000)
    r.set_initial_value(y0, tvals[0]).set_f_params(*args).set_jac_params(*args)
    out = {}
    if trace_until is not None:
        tt, qq = [], []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            while r.t < trace_until:
                r.integrate(trace_until, step=True)
                tt.append(float(r.t))
                qq.append(int(r._integrator.iwork[13]))
        out.update(t=tt, q=qq)
    else:
        ys = [np.array(y0, float)]
        for t in tvals[1:]:
            ys.append(r.integrate(t).copy())
            assert r.successful()
        out.update(y=np.array(ys).tolist())
    iw = r._integrator.iwork
    out.update(nst=int(iw[10]), nfe=int(iw[11]), nje=int(iw[12]), qlast=int(iw[13]), nlu=int(iw[18]),
               nni=int(iw[19]), ncfn=int(iw[20]), netf=int(iw[21]))
    return out


def main():
    d = seir_batch(4)
    C = d["pr"].reshape(4, 4)
    stats = {}
    for b in range(4):
        ps = d["ps"][b]
        args = (ps[0:4], C, ps[4], ps[5])
        stats["seir_batch_%d" % b] = dict(rtol=1e-8, atol=1e-8, tvals=d["tvals"].tolist(), y0=d["y0"][b].tolist(),
                                          ps=ps.tolist(), pr=d["pr"].tolist(), **run(d["y0"][b], d["tvals"], 1e-8, 1e-8, args))
    ps = d["ps"][0]
    args = (ps[0:4], C, ps[4], ps[5])
    stats["seir_trace_T100"] = dict(rtol=1e-8, atol=1e-8, y0=d["y0"][0].tolist(), ps=ps.tolist(), pr=d["pr"].tolist(),
                                    **run(d["y0"][0], np.array([0.0, 100.0]), 1e-8, 1e-8, args, trace_until=100.0))
    with open(os.path.join(GOLD, "dvode_seir.json"), "w") as fh:
        json.dump(stats, fh)
    print({k: tuple(v[c] for c in ("nst", "nfe", "nje", "nlu", "nni", "ncfn", "netf", "qlast")) for k, v in stats.items()})


if __name__ == "__main__":
    main()
