"""Transcendental right-hand sides on the device (pytest -m gpu): the end-to-end parity the rational-only suite lacked.

Models (tools/problems.py): ``forcing`` -- log-parameterised rates, expit switch, logaddexp soft threshold, degree-4
B-spline input through differentiated coefficients (every helper of /root/reference/sunode/symode/lambdify.py:59-77,
275-352; callbacks pinned by reference-generated fixtures); ``logistic_switch`` -- expit / spline of a state, a power
with a differentiated exponent (callbacks pinned by an independent derivation, tests/test_transcendental.py);
``misc`` -- exp / sin / sqrt / log / x^(3/2) / tanh / cos.

Bars: device == oracle BIT FOR BIT (statuses, counters, every fp64 output) -- the generated header embeds
csrc/sa_math.h, so host and device execute one IEEE operation sequence; device vs DOP853 truth at the SURVEY 8(c)
bars (states <= 1e-5, gradients <= 4e-6 relative at rtol = atol = 1e-8).
"""
import os

import numpy as np
import pytest

from tests.helpers import make_oracle, make_problem
from tools.problems import forcing_batch, logistic_switch_batch, misc_batch

pytestmark = pytest.mark.gpu

CMP = [0, 1, 2, 3, 4, 5, 6, 7, 8]
CMP_B = [0, 1, 2, 3, 4, 5, 6, 9, 10, 12]
BATCH = {"forcing": forcing_batch, "logistic_switch": logistic_switch_batch, "misc": misc_batch}
TOL = dict(abstol=1e-8, reltol=1e-8, backward_abstol=1e-8, backward_reltol=1e-8, quad_abstol=1e-8, quad_reltol=1e-8)


def _oracle_run(name, d, grads):
    orc = make_oracle(name)
    cfg = orc.config(rtol=1e-8, atol=1e-8, rtolB=1e-8, atolB=1e-8, rtolQB=1e-8, atolQB=1e-8)
    tv = d["tvals"]
    fwd = orc.solve_forward(cfg, d["y0"], d["ps"], d["pr"], 0.0, tv, nthreads=8)
    bwd = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=8)
    return fwd, bwd


@pytest.mark.parametrize("name", ["mathfn_a", "mathfn_b"])
def test_device_math_library_equals_host_bitwise(name):
    """4 096 random points over wide ranges through the generated callbacks whose outputs are single sa_math.h
    functions (exp, log, log1p, expm1, x^a, x^(-5/3), x^(7/2) | sin, cos, tan, tanh, sinh, cosh) and their
    derivatives: the device's values are the host's, bit for bit."""
    from sunode_amd.solver import Solver
    prob = make_problem(name)
    eng = Solver(prob)._engine()
    orc = make_oracle(name)
    rng = np.random.RandomState(7)
    N = 4096
    y = np.exp(rng.uniform(-6, 4, (N, 5)))
    par = np.exp(rng.uniform(-3, 2, (N, 5))) * rng.choice([-1.0, 1.0], (N, 5))
    lam = rng.randn(N, 5)
    t = rng.uniform(0, 50, N)
    with np.errstate(all="ignore"):
        got = eng.eval_callbacks(t, y, lam, par, np.zeros((N, 0)))
    differing = 0
    for i in range(N):
        host = orc.eval(t[i], y[i], lam[i], par[i], np.zeros(0))
        for key in ("rhs", "jac", "adj", "quad", "adjjac"):
            a, b = np.asarray(got[key][i]).ravel(), np.asarray(host[key]).ravel()
            # (two NaNs count as equal whatever their sign / payload bits)
            differing += int(np.sum((a.view(np.uint64) != b.view(np.uint64)) & ~(np.isnan(a) & np.isnan(b))))
        assert got["codes"][i].tolist() == np.asarray(host["codes"]).tolist()
    assert differing == 0


@pytest.mark.parametrize("name", ["forcing", "logistic_switch", "misc"])
def test_forward_adjoint_bitexact_vs_oracle(name):
    """Forward + adjoint of a transcendental model: statuses, step / order counters and every output equal the
    oracle's bit for bit (B = 300: ragged last wavefront)."""
    from sunode_amd.solver import AdjointSolver, Solver
    prob = make_problem(name)
    d = BATCH[name](300)
    tv = d["tvals"]
    sol = AdjointSolver(prob, **TOL)
    y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], d["ps"], d["pr"])
    g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, d["grads"])
    (yo, so, sto), (go, lo, sbo, stbo) = _oracle_run(name, d, d["grads"])
    assert (st == 0).all() and (stb == 0).all() and (so == 0).all() and (sbo == 0).all()
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(statsb[:, CMP_B], stbo[:, CMP_B])
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)
    # the plain solver (Solver.solve, config-1 path) on the same model
    plain = Solver(prob, abstol=1e-8, reltol=1e-8)
    yp, stp, statsp = plain.solve_batch(0.0, tv, d["y0"], d["ps"], d["pr"])
    orc = make_oracle(name)
    ypo, spo, stpo = orc.solve(orc.config(rtol=1e-8, atol=1e-8), d["y0"], d["ps"], d["pr"], 0.0, tv, nthreads=8)
    np.testing.assert_array_equal(yp, ypo)
    np.testing.assert_array_equal(statsp[:, CMP[:8]], stpo[:, CMP[:8]])


def test_forcing_large_batch_every_instance_equals_the_oracle():
    """8 192 draws of ``forcing`` (128 wavefronts; parameters spread 5x wider than the fixture's): every instance's
    statuses, counters, states and gradients equal the oracle's -- transcendental right-hand sides at batch scale."""
    from sunode_amd.solver import AdjointSolver
    prob = make_problem("forcing")
    B = 8192
    d = forcing_batch(B)
    rng = np.random.RandomState(3)
    ps = d["ps"] + np.array([0.4, 0.4, 0, 0, 0, 0, 0, 0]) * rng.randn(B, 8)      # log_r, log_K over +-1
    tv = d["tvals"]
    sol = AdjointSolver(prob, **TOL)
    y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], ps, d["pr"])
    g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, d["grads"])
    orc = make_oracle("forcing")
    cfg = orc.config(rtol=1e-8, atol=1e-8, rtolB=1e-8, atolB=1e-8, rtolQB=1e-8, atolQB=1e-8)
    yo, so, sto = orc.solve_forward(cfg, d["y0"], ps, d["pr"], 0.0, tv, nthreads=os.cpu_count() or 8)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, d["grads"], nthreads=os.cpu_count() or 8)
    np.testing.assert_array_equal(st, so)
    np.testing.assert_array_equal(stb, sbo)
    assert (st == 0).mean() > 0.99
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(statsb[:, CMP_B], stbo[:, CMP_B])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)


@pytest.mark.parametrize("name", ["forcing", "logistic_switch", "misc"])
def test_forward_adjoint_matches_truth(name, golden_dir):
    """Device vs DOP853 truth (tests/golden/truth_<name>.npz): states <= 1e-5, gradients <= 4e-6 relative."""
    from sunode_amd.solver import AdjointSolver
    d = np.load(os.path.join(golden_dir, "truth_%s.npz" % name))
    sol = AdjointSolver(make_problem(name), **TOL)
    tv = d["tvals"]
    y, st, _ = sol.solve_forward_batch(float(d["t0"]), tv, d["y0"], d["ps"], d["pr"])
    g, lam, stb, _ = sol.solve_backward_batch(tv[-1], float(d["t0"]), tv, d["grads"])
    assert (st == 0).all() and (stb == 0).all()
    assert np.max(np.abs(y - d["y_out"]) / np.abs(d["y_out"]).max(axis=(0, 1))) < 1e-5
    assert np.max(np.abs(g - d["grad_params"]) / np.abs(d["grad_params"]).max(axis=1, keepdims=True)) < 4e-6
    assert np.max(np.abs(-lam - d["grad_y0"]) / np.abs(d["grad_y0"]).max(axis=1, keepdims=True)) < 4e-6


@pytest.mark.parametrize("name,group", [("forcing", "wave4"), ("forcing", "wave16"), ("forcing", "wave"),
                                        ("forcing", "mem"), ("misc", "wave8"), ("logistic_switch", "mem")])
def test_transcendental_model_through_the_other_mappings(name, group, monkeypatch):
    """The callbacks of a transcendental model staged through LDS (lane groups), run by a 4-wavefront workgroup and
    out of the HBM workspace: still the oracle's bits."""
    from sunode_amd.solver import AdjointSolver
    monkeypatch.setenv("SA_FORCE_GROUP", group)
    prob = make_problem(name)
    d = BATCH[name](70)
    tv = d["tvals"]
    sol = AdjointSolver(prob, **TOL)
    y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], d["ps"], d["pr"])
    g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, d["grads"])
    (yo, so, sto), (go, lo, sbo, stbo) = _oracle_run(name, d, d["grads"])
    assert (st == 0).all() and (stb == 0).all()
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(statsb[:, CMP_B], stbo[:, CMP_B])
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)
    sol._engine().close()


@pytest.mark.parametrize("mode", ["simultaneous", "staggered"])
def test_forward_sensitivities_of_a_transcendental_model(mode, golden_dir):
    """``Solver(sens_mode=...)`` on ``forcing`` (8 differentiated parameters x 3 states): sensitivities equal the
    oracle's bit for bit and the chain rule through them reproduces the truth gradient."""
    from sunode_amd.solver import Solver
    prob = make_problem("forcing")
    d = forcing_batch(64)
    tv = d["tvals"]
    sol = Solver(prob, abstol=1e-8, reltol=1e-8, sens_mode=mode)
    sens0 = np.zeros((prob.n_params, prob.n_states))
    y, sens, st, stats = sol.solve_sens_batch(0.0, tv, d["y0"], d["ps"], d["pr"], sens0)
    orc = make_oracle("forcing")
    cfg = orc.config(rtol=1e-8, atol=1e-8)
    yo, seno, so, sto = orc.solve_sens(cfg, d["y0"], d["ps"], d["pr"], sens0, 0.0, tv, mode=mode, nthreads=8)
    assert (st == 0).all() and (so == 0).all()
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(sens, seno)
    np.testing.assert_array_equal(stats[:, CMP[:8]], sto[:, CMP[:8]])
    t = np.load(os.path.join(golden_dir, "truth_forcing.npz"))          # first 8 draws = the truth fixture's
    gp = np.einsum("bki,bkpi->bp", t["grads"], sens[:8])
    assert np.max(np.abs(gp - t["grad_params"]) / np.abs(t["grad_params"]).max(axis=1, keepdims=True)) < 1e-5


def test_nonfinite_transcendental_rhs_is_a_per_instance_failure():
    """log of a state that an instance drives through zero: that instance reports a CVODES failure code with NaN
    outputs, its neighbours integrate (reference: recoverable rhs error, symode/problem.py:266-269)."""
    from sunode_amd.solver import AdjointSolver
    prob = make_problem("misc")
    d = misc_batch(64)
    ps = d["ps"].copy()
    ps[5, 1] = 40.0                     # c[0]: v' = ... - c0 cos(v) drives v negative -> v^(3/2) is NaN
    sol = AdjointSolver(prob, **TOL)
    y, st, _ = sol.solve_forward_batch(0.0, d["tvals"], d["y0"], ps, d["pr"])
    orc = make_oracle("misc")
    cfg = orc.config(rtol=1e-8, atol=1e-8, rtolB=1e-8, atolB=1e-8, rtolQB=1e-8, atolQB=1e-8)
    yo, so, _ = orc.solve_forward(cfg, d["y0"], ps, d["pr"], 0.0, d["tvals"], nthreads=8)
    assert st[5] != 0 and np.isnan(y[5]).any()
    np.testing.assert_array_equal(st, so)
    ok = st == 0
    assert ok.sum() == 63
    np.testing.assert_array_equal(y[ok], yo[ok])
