"""Generate tests/golden/dvode_backward.json: an INDEPENDENT pin of the backward (adjoint) controller.

The forward pins (dvode_stats.json, dvode_seir.json) cover 7 % of the work; the adjoint pass -- restarts, Newton
on -J(y(t))^T, error test, order selection -- was so far only compared HIP-vs-oracle (one author).  Here Fortran
DVODE (scipy.integrate.ode 'vode', method='bdf', analytic Jacobian) integrates

        lambda' = -J(y(t))^T lambda,     lambda(T) = -g,      from T down to t_mid,

where y(t) is the polynomial interpolant of the ORACLE's stored forward points (orc_traj_get: t_k, y_k, order_k)
restated below in numpy (CVApolynomialGetY: Newton divided differences over the last order+1 points), and J is a
plain numpy restatement of the model's Jacobian (not generated code).  The oracle runs the same single interval
with quadrature error control off (CVodeSetQuadErrConB(false): the quadratures then have no say in step / order
selection, so the adjoint system alone is what both codes control) through its normal backward driver:
solve_backward(t0=T, tend=t_mid, tvals=[T]).  t_mid > 0 keeps CVodeB's tstop (= the forward t0) out of play, so the
interval ends CV_NORMAL-style (overshoot + interpolate) exactly like DVODE's itask = 1.

Stored per case: the forward step grid the interpolant was built on (times and orders; the states are the oracle's
forward solution, pinned by the forward fixtures), DVODE's counters (nst, nfe, nlu, nje, nni, ncfn, netf, last order)
per interval end t_mid, lambda(t_mid), and the step trace (t, q per step) of the longest interval.
Robertson: exact agreement holds through the first interval (311 steps); further down its adjoint has an error-test
failure every 7 steps (the interpolant is only C0 at the forward points) and the two codes' round-off drifts apart
like on the forward problem (within a few per cent) -- the test says which rows are exact.
The test (tests/test_oracle_pinning.py) requires the oracle's backward counters to EQUAL DVODE's and lambda(t_mid)
to agree to round-off-level; the GPU test does the same through the device.

    python tools/make_golden_dvode_backward.py
"""
import json
import os
import sys
import warnings

import numpy as np
from scipy.integrate import ode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.make_golden_dvode_seir import seir_j  # noqa: E402
from tools.problems import lv_batch, robertson_batch, seir_batch  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


# ---- CVApolynomialGetY restated (Appendix B of SURVEY.md; oracle/cvodes_oracle.c follows the same digest) ----
class Interpolant:
    def __init__(self, t, y, order):
        self.t, self.y, self.order = np.asarray(t, float), np.asarray(y, float), np.asarray(order, int)
        self._idx, self._tab = None, None

    def _table(self, indx):
        q = int(self.order[indx])
        assert q <= indx
        T = self.t[indx - np.arange(q + 1)]
        Y = self.y[indx - np.arange(q + 1)].copy()
        dt = abs(self.t[indx] - self.t[indx - 1])
        for i in range(1, q + 1):
            for j in range(q, i - 1, -1):
                Y[j] = (dt / (T[j] - T[j - i])) * (Y[j] - Y[j - 1])
        return q, dt, T, Y

    def __call__(self, t):
        indx = int(np.searchsorted(self.t, t, side="left"))       # t[indx-1] < t <= t[indx]
        indx = min(max(indx, 1), len(self.t) - 1)
        if t <= self.t[0]:
            return self.y[0].copy()
        if indx != self._idx:
            self._idx, self._tab = indx, self._table(indx)
        q, dt, T, Y = self._tab
        c, out = 1.0, Y[0].copy()
        for i in range(q):
            c = c * (t - T[i]) / dt
            out = out + c * Y[i + 1]
        return out


# ---- model Jacobians, numpy restatements of tools/problems.py ----
def lv_jac(y, p):
    a, b, c, d = p                   # alpha, beta, gamma, delta:  h' = a h - b l h,  l' = d h l - c l
    h, l = y
    return np.array([[a - b * l, -b * h], [d * l, d * h - c]])


def rob_jac(y, k):
    k1, k2, k3 = k
    y1, y2, y3 = y
    return np.array([[-k1, k2 * y3, k2 * y2],
                     [k1, -k2 * y3 - 2 * k3 * y2, -k2 * y2],
                     [0.0, 2 * k3 * y2, 0.0]])


def dvode_backward(interp, jac, lam_T, T, t_mid, rtol, atol):
    def f(t, lam):
        return -jac(interp(t)).T @ lam

    def jf(t, lam):
        return -jac(interp(t)).T

    r = ode(f, jf).set_integrator("vode", method="bdf", with_jacobian=True, rtol=rtol, atol=atol, nsteps=1000000)
    r.set_initial_value(np.array(lam_T, float), T)
    tt, qq = [], []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        while r.t > t_mid:
            r.integrate(t_mid, step=True)
            assert r.successful()
            tt.append(float(r.t))
            qq.append(int(r._integrator.iwork[13]))
    iw = [int(v) for v in r._integrator.iwork[:22]]
    # same problem once more in normal mode for lambda(t_mid) (the one-step loop above stops past t_mid)
    r2 = ode(f, jf).set_integrator("vode", method="bdf", with_jacobian=True, rtol=rtol, atol=atol, nsteps=1000000)
    r2.set_initial_value(np.array(lam_T, float), T)
    lam_mid = r2.integrate(t_mid)
    assert r2.successful()
    iw2 = r2._integrator.iwork
    stats = dict(nst=int(iw2[10]), nfe=int(iw2[11]), nje=int(iw2[12]), qlast=int(iw2[13]), nlu=int(iw2[18]),
                 nni=int(iw2[19]), ncfn=int(iw2[20]), netf=int(iw2[21]))
    assert stats["nst"] == iw[10] and stats["nfe"] == iw[11], (stats, iw)
    return dict(trace_t=tt, trace_q=qq, lam_mid=lam_mid.tolist(), **stats)


def cases():
    """(tag, problem name, forward tol, y0, ps, pr, full params for the numpy Jacobian, T, [t_mid...], cotangent)"""
    out = []
    d = lv_batch(4)
    for b in range(3):
        p = d["params"][b]
        for tol in ((1e-8, 1e-8), (1e-10, 1e-10)) if b == 0 else ((1e-8, 1e-8),):
            out.append(dict(tag="lv_%d_%g" % (b, tol[0]), problem="lv", rtol=tol[0], atol=tol[1], y0=d["y0"][b], ps=p[:2],
                            pr=p[2:], jac=lambda y, p=p: lv_jac(y, p), T=10.0, t_mids=[9.0, 6.0, 2.5, 0.5],
                            g=np.array([1.0, -0.5])))
    d = robertson_batch(2)
    for b in range(2):
        k = d["params"][b]
        out.append(dict(tag="robertson_%d" % b, problem="robertson", rtol=1e-8, atol=1e-10, y0=d["y0"][b], ps=k,
                        pr=np.zeros(0), jac=lambda y, k=k: rob_jac(y, k), T=40.0, t_mids=[30.0, 4.0],
                        g=np.array([1.0, 1e3, -1.0])))
    d = seir_batch(2)
    C = d["pr"].reshape(4, 4)
    for b in range(2):
        ps = d["ps"][b]
        out.append(dict(tag="seir_%d" % b, problem="seir", rtol=1e-8, atol=1e-8, y0=d["y0"][b], ps=ps, pr=d["pr"],
                        jac=lambda y, ps=ps: seir_j(0.0, y, ps[0:4], C, ps[4], ps[5]), T=100.0,
                        t_mids=[80.0, 30.0, 1.0], g=1.0 + 0.5 * np.cos(0.9 * np.arange(16))))
    return out


def main():
    from oracle.harness import Oracle
    from tests.helpers import make_problem
    gold, summary = {}, {}
    for c in cases():
        prob = make_problem(c["problem"])
        orc = Oracle(prob, c["problem"])
        cfg = orc.config(rtol=c["rtol"], atol=c["atol"], rtolB=c["rtol"], atolB=c["atol"], rtolQB=c["rtol"],
                         atolQB=c["atol"], errconQB=False)
        tv = np.array([c["T"]])
        y, st, _ = orc.solve_forward(cfg, c["y0"][None], c["ps"][None], c["pr"], 0.0, tv)
        assert st[0] == 0
        t_pts, y_pts, q_pts = orc.trajectory(0)
        interp = Interpolant(t_pts, y_pts, q_pts)
        rows = []
        for t_mid in c["t_mids"]:
            dv = dvode_backward(interp, c["jac"], -c["g"], c["T"], t_mid, c["rtol"], c["atol"])
            g, lam, stb, sb = orc.solve_backward(cfg, c["T"], t_mid, tv, c["g"][None, None, :])
            got = [int(v) for v in sb[0][:8]]
            want = [dv[k] for k in ("nst", "nfe", "nlu", "nje", "nni", "ncfn", "netf", "qlast")]
            err = float(np.max(np.abs(lam[0] - dv["lam_mid"]) / np.max(np.abs(dv["lam_mid"]))))
            if t_mid != min(c["t_mids"]):        # the traces of the shorter intervals are prefixes of the longest one's
                dv.pop("trace_t"); dv.pop("trace_q")
            rows.append(dict(t_mid=t_mid, **dv))
            summary["%s@%g" % (c["tag"], t_mid)] = (want, got, "%.1e" % err)
        gold[c["tag"]] = dict(problem=c["problem"], rtol=c["rtol"], atol=c["atol"], y0=np.asarray(c["y0"]).tolist(),
                              ps=np.asarray(c["ps"]).tolist(), pr=np.asarray(c["pr"]).tolist(), T=c["T"],
                              g=c["g"].tolist(), fwd_t=t_pts.tolist(), fwd_q=q_pts.tolist(), intervals=rows)
    for k, v in summary.items():
        print(k, "dvode", v[0], "oracle", v[1], "lam err", v[2], "" if v[0] == v[1] else "   <-- DIFFERENT")
    if "--write" in sys.argv:
        with open(os.path.join(GOLD, "dvode_backward.json"), "w") as fh:
            json.dump(gold, fh)
        print("written")


if __name__ == "__main__":
    main()
