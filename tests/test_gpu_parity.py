"""HIP engine vs CPU oracle on identical inputs (runs on the MI355X box: pytest -m gpu).

Calls go through the C ABI (libsunode_amd.so) via sunode_amd.solver.  Bars:
  * step/order bookkeeping (nst, nfe, nsetups, nje, nni, ncfn, netf, last order, stored
    points, quadrature counters, table rebuilds): bit-exact vs the oracle;
  * states / gradients: bit-exact vs the oracle (both sides evaluate the same IEEE operation
    sequence, -ffp-contract=off, deterministic pow; transcendental right-hand sides through the
    embedded csrc/sa_math.h: tests/test_gpu_transcendental.py);
  * states / gradients vs truth fixtures: within the integration tolerance (written per test).
"""
import os

import numpy as np
import pytest

from tests.helpers import make_oracle, make_problem
from tools.problems import lv_batch, network_batch, robertson_batch, seir_batch

pytestmark = pytest.mark.gpu

CMP = [0, 1, 2, 3, 4, 5, 6, 7, 8]            # nst..qlast, npts
CMP_B = [0, 1, 2, 3, 4, 5, 6, 9, 10, 12]     # backward: + nfqe, netfq, nrebuild


def _lv_inputs(B):
    prob = make_problem("lv")
    d = lv_batch(B)
    ps = d["params"][:, prob.params_subset.subset_index]
    pr = d["params"][:, prob.params_subset.remainder_index]
    return prob, d, ps, pr


def test_device_arithmetic_matches_host_bitwise():
    """IEEE divide / sqrt and the deterministic pow must agree bit-for-bit with the host.  (The math kernel evaluates
    the pow with both coefficient sources -- literals and constant memory, csrc/sa_common.h -- and returns NaN where
    they differ, so this pins both.)"""
    from sunode_amd.solver import Solver
    prob = make_problem("lv")
    eng = Solver(prob)._engine()
    orc = make_oracle("lv")
    rng = np.random.RandomState(1)
    x = np.concatenate([10.0 ** rng.uniform(-300, 300, 20000), 10.0 ** rng.uniform(-12, 3, 40000),
                        [1.0, 6.0, 0.0, -1.0, 5e-324, 1e-310]])
    y = np.concatenate([1.0 / rng.randint(1, 8, 20000), 1.0 / rng.randint(1, 8, 40000), [0.5] * 6])
    # general quotients (mantissas and exponents random within +-1e60): half of them run through the device's
    # unscaled division fdiv (sa_common.h, used by cvSet), all must equal the IEEE quotient
    x = np.concatenate([x, rng.uniform(1, 2, 200000) * 10.0 ** rng.uniform(-60, 60, 200000) * rng.choice([-1, 1], 200000)])
    y = np.concatenate([y, rng.uniform(1, 2, 200000) * 10.0 ** rng.uniform(-60, 60, 200000) * rng.choice([-1, 1], 200000)])
    pw, sq, dv = eng.math_probe(x, y)
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(sq[x >= 0], np.sqrt(x[x >= 0]))
        np.testing.assert_array_equal(dv, x / y)
    k = 60006                                   # det_pow reference (python loop): the first block only
    ref = np.array([orc.det_pow(float(a), float(b)) for a, b in zip(x[:k], y[:k])])
    np.testing.assert_array_equal(pw[:k], ref)


@pytest.mark.parametrize("name", ["lv", "robertson", "seir", "network8", "misc", "notebook", "forcing", "mathfn_a",
                                  "mathfn_b"])
def test_device_callbacks_match_golden(name, golden_dir):
    """Generated device functions vs the reference's own lambdify output (golden vectors)."""
    import json
    from sunode_amd.solver import Solver
    with open(os.path.join(golden_dir, "callbacks.json")) as fh:
        pts = json.load(fh)[name]
    prob = make_problem(name)
    eng = Solver(prob)._engine()
    par = np.array([p["params"] for p in pts])
    got = eng.eval_callbacks([p["t"] for p in pts], [p["y"] for p in pts], [p["lam"] for p in pts],
                             par[:, prob.params_subset.subset_index], par[:, prob.params_subset.remainder_index])
    orc = make_oracle(name)
    for i, p in enumerate(pts):
        host = orc.eval(p["t"], p["y"], p["lam"], par[i, prob.params_subset.subset_index],
                        par[i, prob.params_subset.remainder_index])
        for key in ("rhs", "jac", "adj", "quad", "adjjac"):
            want = np.array(p[key], float).reshape(got[key][i].shape)
            scale = float(np.max(np.abs(want))) if want.size else 0.0
            np.testing.assert_allclose(got[key][i], want, rtol=1e-13, atol=4e-15 * scale)
            # device == host bit for bit: rational right-hand sides AND (round 6) transcendental ones, whose
            # exp / log / sin / pow are the embedded csrc/sa_math.h on both sides
            np.testing.assert_array_equal(got[key][i], host[key].reshape(got[key][i].shape))
        assert got["codes"][i].tolist() == p["codes"]


def test_lv_plain_solve_bitexact_vs_oracle():
    from sunode_amd.solver import Solver
    prob, d, ps, pr = _lv_inputs(512)
    sol = Solver(prob, abstol=1e-8, reltol=1e-8)
    y, status, stats = sol.solve_batch(0.0, d["tvals"], d["y0"], ps, pr)
    orc = make_oracle("lv")
    yo, so, sto = orc.solve(orc.config(rtol=1e-8, atol=1e-8), d["y0"], ps, pr, 0.0, d["tvals"], nthreads=8)
    assert (status == 0).all() and (so == 0).all()
    np.testing.assert_array_equal(stats[:, CMP[:8]], sto[:, CMP[:8]])
    np.testing.assert_array_equal(y, yo)


@pytest.mark.parametrize("tol", [1e-8, 1e-10])
def test_lv_forward_adjoint_bitexact_vs_oracle(tol):
    from sunode_amd.solver import AdjointSolver
    prob, d, ps, pr = _lv_inputs(1024)
    sol = AdjointSolver(prob, abstol=tol, reltol=tol, backward_abstol=tol, backward_reltol=tol,
                        quad_abstol=tol, quad_reltol=tol)
    tv = d["tvals"]
    y, status, stats = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
    grads = np.ones((len(tv), 2))
    g, lam, status_b, stats_b = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    orc = make_oracle("lv")
    cfg = orc.config(rtol=tol, atol=tol, rtolB=tol, atolB=tol, rtolQB=tol, atolQB=tol)
    yo, so, sto = orc.solve_forward(cfg, d["y0"], ps, pr, 0.0, tv, nthreads=8)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=8)
    assert (status == 0).all() and (status_b == 0).all() and (so == 0).all() and (sbo == 0).all()
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(stats_b[:, CMP_B], stbo[:, CMP_B])
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)


def test_lv_gradients_match_truth(golden_dir):
    """rtol=atol=1e-8 everywhere: states <= 1e-5, gradients <= 4e-6 relative to truth
    (north star: gradients within 1e-6 of CVODES, which itself carries ~4e-7 at this tolerance)."""
    from sunode_amd.solver import AdjointSolver
    d = np.load(os.path.join(golden_dir, "truth_lv.npz"))
    prob = make_problem("lv")
    sol = AdjointSolver(prob, abstol=1e-8, reltol=1e-8, backward_abstol=1e-8, backward_reltol=1e-8,
                        quad_abstol=1e-8, quad_reltol=1e-8)
    y, st, _ = sol.solve_forward_batch(float(d["t0"]), d["tvals"], d["y0"], d["ps"], d["pr"])
    g, lam, st2, _ = sol.solve_backward_batch(d["tvals"][-1], float(d["t0"]), d["tvals"], d["grads"])
    assert (st == 0).all() and (st2 == 0).all()
    assert np.max(np.abs(y - d["y_out"]) / np.abs(d["y_out"]).max(axis=(0, 1))) < 1e-5
    assert np.max(np.abs(g - d["grad_params"]) / np.abs(d["grad_params"]).max(axis=1, keepdims=True)) < 4e-6
    assert np.max(np.abs(-lam - d["grad_y0"]) / np.abs(d["grad_y0"]).max(axis=1, keepdims=True)) < 4e-6


def test_robertson_forward_adjoint_vs_oracle_and_truth(golden_dir):
    from sunode_amd.solver import AdjointSolver
    prob = make_problem("robertson")
    d = robertson_batch(256)
    tv = d["tvals"]
    k = np.arange(len(tv))[:, None]; i = np.arange(3)[None, :]
    grads = 1.0 + 0.5 * np.cos(1.7 * k + 0.9 * i)
    sol = AdjointSolver(prob, abstol=1e-10, reltol=1e-8, backward_abstol=1e-10, backward_reltol=1e-8,
                        quad_abstol=1e-10, quad_reltol=1e-8)
    y, status, stats = sol.solve_forward_batch(0.0, tv, d["y0"], d["params"], np.zeros(0))
    g, lam, status_b, stats_b = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    orc = make_oracle("robertson")
    cfg = orc.config(rtol=1e-8, atol=1e-10, rtolB=1e-8, atolB=1e-10, rtolQB=1e-8, atolQB=1e-10)
    yo, so, sto = orc.solve_forward(cfg, d["y0"], d["params"], np.zeros(0), 0.0, tv, nthreads=8)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=8)
    assert (status == 0).all() and (status_b == 0).all()
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(stats_b[:, CMP_B], stbo[:, CMP_B])
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)
    t = np.load(os.path.join(golden_dir, "truth_robertson.npz"))
    assert np.max(np.abs(y[:4] - t["y_out"]) / np.abs(t["y_out"]).max(axis=(0, 1))) < 1e-5
    assert np.max(np.abs(g[:4] - t["grad_params"]) / np.abs(t["grad_params"]).max(axis=1, keepdims=True)) < 1e-5


def test_scalar_api_readme_example():
    """README.md:96-118 through the reference-shaped scalar API (config 1)."""
    from sunode_amd.solver import Solver
    prob = make_problem("lv")
    solver = Solver(prob, sens_mode=None, solver="BDF")
    tvals = np.linspace(0, 10)
    y0 = np.zeros((), dtype=prob.state_dtype)
    y0["hares"] = 1
    y0["lynx"] = 0.1
    solver.set_params_dict({"alpha": 0.1, "beta": 0.2, "gamma": 0.3, "delta": 0.4})
    out = solver.make_output_buffers(tvals)
    solver.solve(t0=0, tvals=tvals, y0=y0, y_out=out)
    np.testing.assert_allclose(out[-1], [1.32497001, 1.04585428], rtol=2e-8)
    assert out.view(prob.state_dtype)["hares"].shape == (50, 1)


@pytest.mark.parametrize("variant", [None, "16", "wave", "mem"])
def test_failures_are_per_instance(variant, monkeypatch):
    """A draw that exhausts the step budget gets CV_TOO_MUCH_WORK + NaN rows; its neighbours finish
    (every kernel family)."""
    from sunode_amd.solver import Solver
    if variant:
        monkeypatch.setenv("SA_FORCE_GROUP", variant)
    prob = make_problem("robertson")
    sol = Solver(prob, abstol=1e-10, reltol=1e-8, mxsteps=100)
    params = np.tile([0.04, 1e4, 3e7], (4, 1))
    params[2] = [0.04, 1e-3, 1e-3]                   # non-stiff draw: few steps
    y, status, stats = sol.solve_batch(0.0, np.array([0.0, 40.0]), np.tile([1.0, 0, 0], (4, 1)), params,
                                       np.zeros(0), max_retries=2)
    assert status.tolist() == [-1, -1, 0, -1]
    assert np.isnan(y[[0, 1, 3]]).all() and np.isfinite(y[2]).all()
    assert (stats[[0, 1, 3], 13] == 2).all()


@pytest.mark.parametrize("variant", ["16", "wave", "mem"])
def test_per_instance_fixed_parameters_and_bad_draws(variant, monkeypatch):
    """SEIR with a different contact matrix per draw (remainder parameters [B][r], rem_stride = r) and
    one draw with a non-finite parameter: the bad draw reports a right-hand-side failure, the others
    match the oracle bit for bit."""
    from sunode_amd.solver import AdjointSolver
    monkeypatch.setenv("SA_FORCE_GROUP", variant)
    prob = make_problem("seir")
    B = 9
    d = seir_batch(B)
    rng = np.random.RandomState(5)
    pr = d["pr"][None, :] * np.exp(0.1 * rng.randn(B, 16))
    ps = d["ps"].copy()
    ps[4, 0] = np.nan
    tv = d["tvals"][:11]
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(16)[None, :])
    tol = 1e-8
    sol = AdjointSolver(prob, abstol=tol, reltol=tol, backward_abstol=tol, backward_reltol=tol,
                        quad_abstol=tol, quad_reltol=tol, max_steps=1024)
    y, status, stats = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
    g, lam, status_b, stats_b = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    orc = make_oracle("seir")
    cfg = orc.config(rtol=tol, atol=tol, rtolB=tol, atolB=tol, rtolQB=tol, atolQB=tol)
    yo, so, sto = orc.solve_forward(cfg, d["y0"], ps, pr, 0.0, tv, nthreads=4)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=4)
    assert status.tolist() == so.tolist() and status_b.tolist() == sbo.tolist()
    assert status[4] != 0 and status_b[4] != 0 and (np.delete(status, 4) == 0).all()
    assert np.isnan(y[4]).all() and np.isnan(g[4]).all()
    ok = np.arange(B) != 4
    np.testing.assert_array_equal(y[ok], yo[ok])
    np.testing.assert_array_equal(g[ok], go[ok])
    np.testing.assert_array_equal(lam[ok], lo[ok])
    np.testing.assert_array_equal(stats[ok][:, CMP], sto[ok][:, CMP])


def test_seir_forward_adjoint_vs_oracle_and_truth(golden_dir):
    """Config 4 problem (16 states, 8 differentiated + 16 shared fixed parameters)."""
    from sunode_amd.solver import AdjointSolver
    prob = make_problem("seir")
    d = seir_batch(128)
    tv = d["tvals"]
    k = np.arange(len(tv))[:, None]; i = np.arange(16)[None, :]
    grads = 1.0 + 0.5 * np.cos(1.7 * k + 0.9 * i)
    tol = 1e-8
    sol = AdjointSolver(prob, abstol=tol, reltol=tol, backward_abstol=tol, backward_reltol=tol,
                        quad_abstol=tol, quad_reltol=tol, max_steps=1024)
    y, status, stats = sol.solve_forward_batch(0.0, tv, d["y0"], d["ps"], d["pr"])
    g, lam, status_b, stats_b = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    orc = make_oracle("seir")
    cfg = orc.config(rtol=tol, atol=tol, rtolB=tol, atolB=tol, rtolQB=tol, atolQB=tol)
    yo, so, sto = orc.solve_forward(cfg, d["y0"], d["ps"], d["pr"], 0.0, tv, nthreads=8)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=8)
    assert (status == 0).all() and (status_b == 0).all()
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(stats_b[:, CMP_B], stbo[:, CMP_B])
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)
    t = np.load(os.path.join(golden_dir, "truth_seir.npz"))
    assert np.max(np.abs(y[:2] - t["y_out"]) / np.abs(t["y_out"]).max(axis=(0, 1))) < 1e-5
    scale = np.abs(t["grad_params"]).max(axis=1, keepdims=True)
    assert np.max(np.abs(g[:2] - t["grad_params"]) / scale) < 3e-5


@pytest.mark.parametrize("name,group", [("lv", 8), ("notebook", 8), ("robertson", 16),
                                        ("lv", "mem"), ("notebook", "mem"), ("robertson", "mem"),
                                        ("lv", "wave"), ("notebook", "wave"), ("robertson", "wave"),
                                        ("lv", "wave8"), ("robertson", "wave16")])
def test_cooperative_mapping_equals_thread_per_instance(name, group, monkeypatch):
    """The same problem through the kernel families (SA_FORCE_GROUP): G lanes per instance with
    butterfly norms and row-distributed LU, the wavefront-per-instance kernel with the LDS-resident
    LU ("wave", 64 < n <= 128) or the memory-resident generic kernel ("mem", n > 128) must
    reproduce the one-lane-per-instance register kernel bit for bit (and therefore the oracle)."""
    from sunode_amd.solver import AdjointSolver
    prob = make_problem(name)
    rng = np.random.RandomState(3)
    B = 100                                            # ragged vs 64/G instances per wavefront
    if name == "lv":
        d = lv_batch(B)
        y0, ps, pr, tv = d["y0"], d["params"][:, :2], d["params"][:, 2:], d["tvals"]
        kw = dict(abstol=1e-8, reltol=1e-8, backward_abstol=1e-8, backward_reltol=1e-8, quad_abstol=1e-8,
                  quad_reltol=1e-8)
    elif name == "robertson":
        d = robertson_batch(B)
        y0, ps, pr, tv = d["y0"], d["params"], np.zeros(0), d["tvals"]
        kw = dict(abstol=1e-10, reltol=1e-8, backward_abstol=1e-10, backward_reltol=1e-8, quad_abstol=1e-10,
                  quad_reltol=1e-8)
    else:
        y0 = np.abs(rng.randn(B, 5)) + 0.5
        ps = rng.randn(B, 3)
        pr = np.linspace(0, 1, 50)
        tv = np.arange(20) / 100
        kw = {}
    n = prob.n_states
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(n)[None, :])
    results = []
    for g in ("1", str(group)):
        monkeypatch.setenv("SA_FORCE_GROUP", g)
        sol = AdjointSolver(prob, **kw)
        y, st, stats = sol.solve_forward_batch(float(tv[0]), tv, y0, ps, pr)
        gr, lam, stb, statsb = sol.solve_backward_batch(tv[-1], float(tv[0]), tv, grads)
        assert (st == 0).all() and (stb == 0).all()
        from sunode_amd.solver import Solver
        plain = Solver(prob, **{k: v for k, v in kw.items() if k in ("abstol", "reltol")})
        yp, stp, _ = plain.solve_batch(float(tv[0]), tv, y0, ps, pr)
        results.append((y, stats[:, CMP], gr, lam, statsb[:, CMP_B], yp, stp))
        sol._engine().close()
        plain._engine().close()
    for a, b in zip(*results):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("variant", ["16", "mem", "wave", "wave4", "wave16", "wave32"])
def test_seir_large_system_mappings_vs_oracle(variant, monkeypatch):
    """SEIR (n = 16, 16 shared fixed parameters; the engine's own choice is bdf_wave.hip with 8 lanes per
    instance) through the other mappings: 16 lanes per instance, memory-resident, wavefront-per-instance,
    4 / 16 / 32 lanes per instance: all bit-exact against the oracle."""
    from sunode_amd.solver import AdjointSolver
    monkeypatch.setenv("SA_FORCE_GROUP", variant)
    prob = make_problem("seir")
    d = seir_batch(70)
    tv = d["tvals"]
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(16)[None, :])
    tol = 1e-8
    sol = AdjointSolver(prob, abstol=tol, reltol=tol, backward_abstol=tol, backward_reltol=tol,
                        quad_abstol=tol, quad_reltol=tol, max_steps=1024)
    y, status, stats = sol.solve_forward_batch(0.0, tv, d["y0"], d["ps"], d["pr"])
    g, lam, status_b, stats_b = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    orc = make_oracle("seir")
    cfg = orc.config(rtol=tol, atol=tol, rtolB=tol, atolB=tol, rtolQB=tol, atolQB=tol)
    yo, so, sto = orc.solve_forward(cfg, d["y0"], d["ps"], d["pr"], 0.0, tv, nthreads=8)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=8)
    assert (status == 0).all() and (status_b == 0).all()
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(stats_b[:, CMP_B], stbo[:, CMP_B])
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)


@pytest.mark.parametrize("variant", ["wave", "mem"])
def test_network100_forward_adjoint_vs_oracle(variant, monkeypatch):
    """Config 5 (BASELINE.json): 100 states, shared fixed 100x100 rate matrix, 4 differentiated
    parameters.  Between 65 and 128 states the engine selects the wavefront-per-instance kernel
    by itself; the memory-resident kernel (the mapping above 128 states) is forced on the same
    problem over a shorter horizon."""
    from sunode_amd import _native
    from sunode_amd.solver import AdjointSolver
    prob = make_problem("network100")
    assert _native.kernel_variant(prob.native_source())[0] == "bdf_wave.hip"
    if variant == "mem":
        monkeypatch.setenv("SA_FORCE_GROUP", "mem")
    B = 6
    d = network_batch(B)
    tv = d["tvals"] if variant == "wave" else d["tvals"][:3]
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(100)[None, :])
    tol = 1e-8
    sol = AdjointSolver(prob, abstol=tol, reltol=tol, backward_abstol=tol, backward_reltol=tol,
                        quad_abstol=tol, quad_reltol=tol, max_steps=1024)
    y, status, stats = sol.solve_forward_batch(0.0, tv, d["y0"], d["ps"], d["pr"])
    g, lam, status_b, stats_b = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    orc = make_oracle("network100")
    cfg = orc.config(rtol=tol, atol=tol, rtolB=tol, atolB=tol, rtolQB=tol, atolQB=tol)
    yo, so, sto = orc.solve_forward(cfg, d["y0"], d["ps"], d["pr"], 0.0, tv, nthreads=6)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=6)
    assert (status == 0).all() and (status_b == 0).all() and (so == 0).all() and (sbo == 0).all()
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(stats_b[:, CMP_B], stbo[:, CMP_B])
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)


@pytest.mark.parametrize("variant", [None, "wave8", "wave32", "wave", "mem"])
def test_matvec_callbacks_in_every_mapping(variant, monkeypatch):
    """network24: the right-hand side and the adjoint right-hand side contain a dense 24 x 24 fixed-parameter block
    that is generated as one lane-parallel matrix-vector product (SA_MATVEC).  Lane groups of 8 / 16 (engine
    choice) / 32, the workgroup-per-instance build (matrix-vector split over four wavefronts, register-resident
    LU) and the memory-resident build must all reproduce the oracle bit for bit."""
    from sunode_amd import _native
    from sunode_amd.solver import AdjointSolver
    prob = make_problem("network24")
    assert _native.kernel_variant(prob.native_source()) == ("bdf_wave.hip", 16)
    if variant:
        monkeypatch.setenv("SA_FORCE_GROUP", variant)
    B, n = 40, 24
    d = network_batch(B, n=n)
    tv = d["tvals"][:6]
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(n)[None, :])
    tol = 1e-8
    sol = AdjointSolver(prob, abstol=tol, reltol=tol, backward_abstol=tol, backward_reltol=tol,
                        quad_abstol=tol, quad_reltol=tol)
    eng = sol._engine()
    rng = np.random.RandomState(5)
    pts = 7
    tt = np.linspace(0, 1, pts)
    yy = d["y0"][:pts] * np.exp(0.1 * rng.randn(pts, n))
    ll = rng.randn(pts, n)
    pr_native = prob.extend_remainder(d["pr"])
    ev = eng.eval_callbacks(tt, yy, ll, d["ps"][:pts], np.tile(pr_native, (pts, 1)))
    orc = make_oracle("network24")
    for i in range(pts):
        host = orc.eval(tt[i], yy[i], ll[i], d["ps"][i], pr_native)
        for key in ("rhs", "adj", "quad", "jac", "adjjac"):
            np.testing.assert_array_equal(ev[key][i], host[key].reshape(ev[key][i].shape))
    y, status, stats = sol.solve_forward_batch(0.0, tv, d["y0"], d["ps"], d["pr"])
    g, lam, status_b, stats_b = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    cfg = orc.config(rtol=tol, atol=tol, rtolB=tol, atolB=tol, rtolQB=tol, atolQB=tol)
    yo, so, sto = orc.solve_forward(cfg, d["y0"], d["ps"], d["pr"], 0.0, tv, nthreads=8)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=8)
    assert (status == 0).all() and (status_b == 0).all() and (so == 0).all() and (sbo == 0).all()
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(stats_b[:, CMP_B], stbo[:, CMP_B])
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)


@pytest.mark.parametrize("variant", [None, "16", "wave", "mem"])
def test_lamda_all_out_and_quad_all_out(variant, monkeypatch):
    """Optional per-output-time results of solve_backward (reference solver.py:778-781) from every
    kernel family, bit-for-bit against the oracle; scalar API fills caller-allocated arrays."""
    from sunode_amd.solver import AdjointSolver
    if variant:
        monkeypatch.setenv("SA_FORCE_GROUP", variant)
    prob = make_problem("robertson")
    d = robertson_batch(5)
    tv = d["tvals"]
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(3)[None, :])
    kw = dict(abstol=1e-10, reltol=1e-8, backward_abstol=1e-10, backward_reltol=1e-8, quad_abstol=1e-10,
              quad_reltol=1e-8)
    sol = AdjointSolver(prob, **kw)
    sol.solve_forward_batch(0.0, tv, d["y0"], d["params"], np.zeros(0))
    g, lam, st, _, lam_all, quad_all = sol.solve_backward_batch(tv[-1], 0.0, tv, grads, return_all=True)
    orc = make_oracle("robertson")
    cfg = orc.config(rtol=1e-8, atol=1e-10, rtolB=1e-8, atolB=1e-10, rtolQB=1e-8, atolQB=1e-10)
    orc.solve_forward(cfg, d["y0"], d["params"], np.zeros(0), 0.0, tv)
    go, lo, so, _, lao, qao = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, return_all=True)
    assert (st == 0).all() and (so == 0).all()
    np.testing.assert_array_equal(lam_all, lao)
    np.testing.assert_array_equal(quad_all, qao)
    np.testing.assert_array_equal(g, go)
    if variant is None:
        sol.set_params_dict({"k1": float(d["params"][0, 0]), "k2": float(d["params"][0, 1]),
                             "k3": float(d["params"][0, 2])})
        y_out, grad_out, lamda_out = sol.make_output_buffers(tv)
        la = np.zeros((len(tv), 3)); qa = np.zeros((len(tv), 3))
        sol.solve_forward(0.0, tv, d["y0"][0], y_out)
        sol.solve_backward(tv[-1], 0.0, tv, grads, grad_out, lamda_out, lamda_all_out=la, quad_all_out=qa)
        np.testing.assert_array_equal(la, lao[0])
        np.testing.assert_array_equal(qa, qao[0])


@pytest.mark.parametrize("variant", [None, "16", "wave", "mem"])
def test_inequality_constraints_match_oracle(variant, monkeypatch):
    """Solver / AdjointSolver(constraints=...) (reference solver.py:230-233, 569-572): the constraint build of
    every kernel family follows the oracle bit for bit, including draws that end in CV_CONSTR_FAIL / CV_ILL_INPUT."""
    from sunode_amd.solver import AdjointSolver, Solver
    if variant:
        monkeypatch.setenv("SA_FORCE_GROUP", variant)
    prob = make_problem("robertson")
    B = 6
    d = robertson_batch(B)
    tv = np.array([0.0] + [4.0 * 10.0 ** k for k in range(11)])
    y0 = d["y0"].copy()
    y0[3] = [0.0, 0.5, 0.5]                                   # violates y1 > 0 at t0
    cons = np.array([2.0, 1.0, 1.0])
    sol = Solver(prob, abstol=1e-7, reltol=1e-4, constraints=cons, mxsteps=5000)
    y, st, stats = sol.solve_batch(0.0, tv, y0, d["params"], np.zeros(0))
    orc = make_oracle("robertson")
    cfg = orc.config(rtol=1e-4, atol=1e-7, mxstep=5000, constraints=cons)
    yo, so, sto = orc.solve(cfg, y0, d["params"], np.zeros(0), 0.0, tv)
    assert st.tolist() == so.tolist() and st[3] == -22 and (np.delete(st, 3) == 0).any()
    ok = st == 0
    np.testing.assert_array_equal(y[ok], yo[ok])
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    assert np.nanmin(y[ok]) >= 0.0
    # unconstrained, the same draws leave the physical region
    free = Solver(prob, abstol=1e-7, reltol=1e-4, mxsteps=5000)
    yf, stf, _ = free.solve_batch(0.0, tv, d["y0"], d["params"], np.zeros(0))
    assert np.nanmin(yf) < -1.0
    # the adjoint solver applies the constraints to its forward pass
    adj = AdjointSolver(prob, abstol=1e-7, reltol=1e-4, constraints=cons, mxsteps=5000, max_steps=4096)
    ya, sta, statsa = adj.solve_forward_batch(0.0, tv[:8], y0, d["params"], np.zeros(0))
    yao, sao, statsao = orc.solve_forward(orc.config(rtol=1e-4, atol=1e-7, mxstep=5000, constraints=cons),
                                          y0, d["params"], np.zeros(0), 0.0, tv[:8])
    assert sta.tolist() == sao.tolist()
    np.testing.assert_array_equal(ya[sta == 0], yao[sao == 0])


@pytest.mark.parametrize("name,variant", [("lv", None), ("lv", "8"), ("robertson", None), ("robertson", "16"),
                                          ("seir", None), ("seir", "wave"), ("seir", "wave16"), ("seir", "mem")])
def test_hermite_interpolation_matches_oracle(name, variant, monkeypatch):
    """AdjointSolver(interpolation='hermite') (reference solver.py:581-582): Hermite builds of the register
    (default for small systems), lane-group, workgroup and memory kernels against the oracle."""
    from sunode_amd import _native
    from sunode_amd.solver import AdjointSolver
    if variant:
        monkeypatch.setenv("SA_FORCE_GROUP", variant)
    prob = make_problem(name)
    B = 37
    if name == "lv":
        d = lv_batch(B); ps, pr = d["params"][:, :2], d["params"][:, 2:]; rt, at = 1e-8, 1e-8
        assert _native.kernel_variant(prob.native_source(), hermite=True) == \
            (("bdf_wave.hip", 8) if variant == "8" else ("bdf_kernels.hip", 1))
    elif name == "robertson":
        d = robertson_batch(B); ps, pr = d["params"], np.zeros(0); rt, at = 1e-8, 1e-10
    else:
        B = 20
        d = seir_batch(B); ps, pr = d["ps"], d["pr"]; rt, at = 1e-8, 1e-8
    tv = d["tvals"]
    n = prob.n_states
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(n)[None, :])
    sol = AdjointSolver(prob, abstol=at, reltol=rt, backward_abstol=at, backward_reltol=rt, quad_abstol=at,
                        quad_reltol=rt, interpolation="hermite", max_steps=2048)
    y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
    g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    orc = make_oracle(name)
    cfg = orc.config(rtol=rt, atol=at, rtolB=rt, atolB=at, rtolQB=rt, atolQB=at, hermite=True)
    yo, so, sto = orc.solve_forward(cfg, d["y0"], ps, pr, 0.0, tv, nthreads=8)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=8)
    assert (st == 0).all() and (stb == 0).all() and (so == 0).all() and (sbo == 0).all()
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(stats_b_cols(statsb), stats_b_cols(stbo))
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)


def stats_b_cols(stats):
    return stats[:, CMP_B]


_TORCH_SCRIPT = r"""
import sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
from sunode_amd import _native
from sunode_amd.solver import AdjointSolver
from tests.helpers import make_problem
from tools.problems import lv_batch

prob = make_problem("lv")
d = lv_batch(300)
ps = d["params"][:, prob.params_subset.subset_index]
pr = d["params"][:, prob.params_subset.remainder_index]
dev = torch.device("cuda", 0)
tol = 1e-8
tv = d["tvals"]
n_t = len(tv)
t = {k: torch.tensor(np.ascontiguousarray(v), device=dev) for k, v in dict(ps=ps, pr=pr, y0=d["y0"], tvals=tv).items()}
grads = torch.ones((n_t, 2), dtype=torch.float64, device=dev)
y_out = torch.empty((300, n_t, 2), dtype=torch.float64, device=dev)
g_out = torch.empty((300, 2), dtype=torch.float64, device=dev)
l_out = torch.empty((300, 2), dtype=torch.float64, device=dev)
st_f = torch.empty(300, dtype=torch.int32, device=dev); st_b = torch.empty(300, dtype=torch.int32, device=dev)
sf = torch.empty((300, 16), dtype=torch.int64, device=dev); sb = torch.empty((300, 16), dtype=torch.int64, device=dev)
eng = _native.NativeSolver(prob.native_source(), device=0, rtol=tol, atol=tol, rtolB=tol, atolB=tol,
                           rtolQB=tol, atolQB=tol, traj_capacity=512, n_states=2)
torch.cuda.synchronize()                          # raises if solver creation left an error behind
eng.solve(_native.SA_MEM_DEVICE, 300, t["y0"], t["ps"], t["pr"], 2, 0.0, t["tvals"], n_t, y_out, st_f, sf, adjoint=True)
eng.solve_backward(_native.SA_MEM_DEVICE, 300, t["ps"], t["pr"], 2, float(tv[-1]), 0.0, t["tvals"], n_t, grads, 0,
                   g_out, l_out, st_b, sb)
eng.synchronize()
assert int((st_f != 0).sum().item()) == 0 and int((st_b != 0).sum().item()) == 0
assert np.isfinite(float((y_out.sum() + g_out.sum()).item()))
torch.cuda.synchronize()
host = AdjointSolver(prob, abstol=tol, reltol=tol, backward_abstol=tol, backward_reltol=tol, quad_abstol=tol,
                     quad_reltol=tol, max_steps=512)
y, _, _ = host.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
g, lam, _, _ = host.solve_backward_batch(tv[-1], 0.0, tv, np.ones((n_t, 2)))
assert np.array_equal(y_out.cpu().numpy(), y)
assert np.array_equal(g_out.cpu().numpy(), g) and np.array_equal(l_out.cpu().numpy(), lam)
print("TORCH_PATH_OK")
"""


def test_device_resident_buffers_with_torch():
    """The path bench.py times: torch-ROCm tensors in, tensors out (SA_MEM_DEVICE), PyTorch kernels on the
    same device before and after -- in a fresh process, like bench.py.  The library must not leave a HIP error
    behind for PyTorch to trip over, and the results must equal the host-buffer path."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", _TORCH_SCRIPT, root], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "TORCH_PATH_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


def _check_call_solve(solver, params, deriv):
    """The reference's own helper (sunode/test_solve.py:81-117), reference call shapes."""
    solver.set_params_dict(params)
    time = np.linspace(0, 1)
    if deriv == "forward":
        y_buffer, sense_buffer = solver.make_output_buffers(time)
        solver.solve(0, time, np.ones_like(y_buffer[0]), y_buffer, sens0=np.zeros_like(sense_buffer[0]),
                     sens_out=sense_buffer)
        return time, y_buffer, sense_buffer
    if deriv == "backward":
        y_buffer, grads_buffer, lamda_buffer = solver.make_output_buffers(time)
        solver.solve_forward(0, time, np.ones_like(y_buffer[0]), y_buffer)
        grads = np.ones((len(time), y_buffer.shape[-1]))
        solver.solve_backward(time[-1], time[0], time, grads, grads_buffer, lamda_buffer)
        return time, y_buffer, grads_buffer, lamda_buffer
    y_buffer = solver.make_output_buffers(time)
    solver.solve(0, time, np.ones_like(y_buffer[0]), y_buffer)
    return time, y_buffer


def test_reference_test_declare_sens_and_linear_solver_kwarg():
    """sunode/test_solve.py:120-181 run against the device engine, plus the analytic answers the reference
    does not check: x' = x + b, x(0) = 1  ->  x = (1 + b) e^t - b, dx/db = e^t - 1, dL/db = sum_k (e^{t_k} - 1)."""
    import warnings
    from sunode_amd import SympyProblem
    from sunode_amd.solver import AdjointSolver, Solver

    def rhs(t, y, p):
        return {"x": y.x + p.a.b}

    problem = SympyProblem({"a": {"b": ()}}, {"x": ()}, rhs, derivative_params=[("a", "b")])
    vals = {"a": {"b": 0.2}}
    for mode in ("simultaneous", "staggered"):
        t, y, sens = _check_call_solve(Solver(problem, sens_mode=mode), vals, "forward")
        np.testing.assert_allclose(y[:, 0], 1.2 * np.exp(t) - 0.2, rtol=1e-8)
        np.testing.assert_allclose(sens[:, 0, 0], np.exp(t) - 1.0, rtol=1e-7, atol=1e-9)
    t, y = _check_call_solve(Solver(problem), vals, None)
    np.testing.assert_allclose(y[:, 0], 1.2 * np.exp(t) - 0.2, rtol=1e-8)
    t, y, g, lam = _check_call_solve(AdjointSolver(problem), vals, "backward")
    np.testing.assert_allclose(g[0], np.sum(np.exp(t) - 1.0), rtol=1e-7)
    np.testing.assert_allclose(-lam[0], np.sum(np.exp(t)), rtol=1e-7)         # dL/dx(0)

    problem2 = SympyProblem({"b": ()}, {"x": ()}, lambda t, y, p: {"x": y.x}, derivative_params=[])
    for linear_solver in ["dense", "dense_finitediff", "spgmr_finitediff", "spgmr", "band"]:
        kw = {"upper_bandwidth": 1, "lower_bandwidth": 1} if linear_solver == "band" else {}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            solver = Solver(problem2, linear_solver=linear_solver, linear_solver_kwargs=kw)
        t, y = _check_call_solve(solver, {"b": 0.2}, None)
        np.testing.assert_allclose(y[:, 0], np.exp(t), rtol=1e-8)


@pytest.mark.parametrize("variant", [None, "wave2", "wave4", "wave8", "wave16", "wave", "mem"])
def test_row_exchanges_in_the_dense_lu(variant, monkeypatch):
    """A system whose Newton matrix is far from diagonally dominant (rotations at 1000 rad/s far below the
    tolerances, steps of order 1): the partial-pivoting LU has to exchange rows in the forward and in the
    backward solve.  Every mapping (lean lane groups by default, other group sizes, workgroup, memory-resident) must
    follow the oracle's pivot choices bit for bit."""
    from sunode_amd.solver import AdjointSolver
    if variant:
        monkeypatch.setenv("SA_FORCE_GROUP", variant)
    prob = make_problem("pivoting")
    B = 21
    rng = np.random.RandomState(0)
    ps = np.array([0.5, 0.3]) * np.exp(0.1 * rng.randn(B, 2))
    pr = np.array([1000.0, 700.0])
    y0 = np.tile([0.5, 1e-6, 0.0, 1e-6, 2e-6, 0.1], (B, 1))
    tv = np.linspace(0, 20, 11)
    grads = np.ones((11, 6)); grads[:, 1:5] = 0.0
    kw = dict(abstol=1e-4, reltol=1e-5, backward_abstol=1e-4, backward_reltol=1e-5, quad_abstol=1e-4,
              quad_reltol=1e-5)
    sol = AdjointSolver(prob, **kw)
    y, st, stats = sol.solve_forward_batch(0.0, tv, y0, ps, pr)
    g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    orc = make_oracle("pivoting")
    cfg = orc.config(rtol=1e-5, atol=1e-4, rtolB=1e-5, atolB=1e-4, rtolQB=1e-5, atolQB=1e-4)
    yo, so, sto = orc.solve_forward(cfg, y0, ps, pr, 0.0, tv, nthreads=4)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=4)
    assert (st == 0).all() and (stb == 0).all()
    assert stats[:, 0].max() < 80                      # steps of order 0.5: gamma * omega >> 1, rows get exchanged
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(statsb[:, CMP_B], stbo[:, CMP_B])
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)


@pytest.mark.parametrize("variant", [None, "wave"])
def test_pivots_beyond_the_lean_reciprocal_range(variant, monkeypatch):
    """Diagonal entries of the Newton matrix around 1e200 (exactly-zero components with decay rates of 1e200, steps of
    order 1): the workgroup LU's speculative panel factorisation computes reciprocals as v_rcp_f64 + 6 FMA, which
    equals the IEEE division only while nothing is scaled -- pivots beyond 2^500 must be sent to the general code
    (exponent guard) and the factors must still equal the oracle's bit for bit."""
    from sunode_amd.solver import AdjointSolver
    if variant:
        monkeypatch.setenv("SA_FORCE_GROUP", variant)
    prob = make_problem("huge_pivots")
    B = 9
    rng = np.random.RandomState(1)
    ps = np.array([0.5, 0.3]) * np.exp(0.1 * rng.randn(B, 2))
    pr = np.array([1e200, 3e199])
    y0 = np.tile([0.5, 0.0, 0.0, 0.0, 0.0, 0.1], (B, 1))
    tv = np.linspace(0, 10, 6)
    grads = np.ones((6, 6)); grads[:, 1:5] = 0.0
    kw = dict(abstol=1e-6, reltol=1e-6, backward_abstol=1e-6, backward_reltol=1e-6, quad_abstol=1e-6, quad_reltol=1e-6)
    sol = AdjointSolver(prob, **kw)
    y, st, stats = sol.solve_forward_batch(0.0, tv, y0, ps, pr)
    g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    orc = make_oracle("huge_pivots")
    cfg = orc.config(rtol=1e-6, atol=1e-6, rtolB=1e-6, atolB=1e-6, rtolQB=1e-6, atolQB=1e-6)
    yo, so, sto = orc.solve_forward(cfg, y0, ps, pr, 0.0, tv)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads)
    assert (so == 0).all() and (sbo == 0).all() and (st == 0).all() and (stb == 0).all()
    assert (yo[:, :, 1:5] == 0.0).all()                 # the stiff components never move: steps are set by the slow ones
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(statsb[:, CMP_B], stbo[:, CMP_B])
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)


@pytest.mark.parametrize("variant", [None, "8", "wave", "mem"])
def test_recoverable_rhs_failures_match_oracle(variant, monkeypatch):
    """x' = -k sqrt(x): draws that reach x = 0 inside the horizon hit a non-finite right-hand side; CVODES treats
    that as a recoverable failure (step cut by 4, up to 10 times), then gives up with CV_REPTD_RHSFUNC_ERR.  Same
    counters, same status, NaN rows, and CV_NO_FWD from the backward pass -- in every mapping."""
    from sunode_amd.solver import AdjointSolver
    if variant:
        monkeypatch.setenv("SA_FORCE_GROUP", variant)
    prob = make_problem("sqrt_decay")
    ps = np.array([[1.0], [1.2], [0.5], [2.0], [0.7]])
    y0 = np.tile([1.0, 1.0], (5, 1))
    tv = np.linspace(0, 2.5, 6)
    kw = dict(abstol=1e-10, reltol=1e-8, backward_abstol=1e-10, backward_reltol=1e-8, quad_abstol=1e-10,
              quad_reltol=1e-8)
    sol = AdjointSolver(prob, **kw)
    y, st, stats = sol.solve_forward_batch(0.0, tv, y0, ps, np.zeros(0))
    g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, np.ones((6, 2)))
    orc = make_oracle("sqrt_decay")
    cfg = orc.config(rtol=1e-8, atol=1e-10, rtolB=1e-8, atolB=1e-10, rtolQB=1e-8, atolQB=1e-10)
    yo, so, sto = orc.solve_forward(cfg, y0, ps, np.zeros(0), 0.0, tv)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, np.ones((6, 2)))
    assert st.tolist() == so.tolist() == [-10, -10, 0, -10, 0]
    assert stb.tolist() == sbo.tolist() == [-102, -102, 0, -102, 0]
    np.testing.assert_array_equal(stats[:, CMP[:8]], sto[:, CMP[:8]])
    ok = st == 0
    np.testing.assert_array_equal(y[ok], yo[ok])
    np.testing.assert_array_equal(g[ok], go[ok])
    assert np.isnan(y[~ok]).all() and np.isnan(g[~ok]).all() and np.isnan(lam[~ok]).all()
    np.testing.assert_allclose(y[2, :, 0], (1 - 0.25 * tv) ** 2, rtol=1e-7)         # x = (1 - k t / 2)^2


@pytest.mark.parametrize("variant", [None, "8", "wave4", "wave", "mem"])
def test_error_test_failure_paths_match_oracle(variant, monkeypatch):
    """Discontinuous forcing (two Heaviside switches of growing size): dozens of error-test failures per solve,
    repeated failures inside one step (order reduction, reload of the derivative column at order 1), forward and
    backward.  Counters (incl. netf / netfQ) and results bit for bit in every mapping."""
    from sunode_amd.solver import AdjointSolver
    if variant:
        monkeypatch.setenv("SA_FORCE_GROUP", variant)
    prob = make_problem("switched")
    ps = np.array([[1.0, 50.0], [2.0, 500.0], [0.5, 5000.0], [3.0, 1e5], [1.5, 1e7]])
    y0 = np.tile([1.0, 0.0], (5, 1))
    tv = np.linspace(0, 3, 7)
    grads = np.ones((7, 2))
    kw = dict(abstol=1e-10, reltol=1e-8, backward_abstol=1e-10, backward_reltol=1e-8, quad_abstol=1e-10,
              quad_reltol=1e-8)
    sol = AdjointSolver(prob, **kw)
    y, st, stats = sol.solve_forward_batch(0.0, tv, y0, ps, np.zeros(0))
    g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    orc = make_oracle("switched")
    cfg = orc.config(rtol=1e-8, atol=1e-10, rtolB=1e-8, atolB=1e-10, rtolQB=1e-8, atolQB=1e-10)
    yo, so, sto = orc.solve_forward(cfg, y0, ps, np.zeros(0), 0.0, tv)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads)
    assert st.tolist() == so.tolist() and stb.tolist() == sbo.tolist() and (st == 0).all() and (stb == 0).all()
    assert (stats[:, 6] >= 40).all()                      # netf: the switches are hit hard
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(statsb[:, CMP_B], stbo[:, CMP_B])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)


@pytest.mark.parametrize("variant", [None, "16", "wave", "mem"])
def test_vector_abstol_matches_oracle(variant, monkeypatch):
    """Per-component absolute tolerances (reference solver.py:404-407, 624-635: CVodeSVtolerances) in the
    forward problem, scalar ones in the backward problem."""
    from sunode_amd.solver import AdjointSolver, Solver
    if variant:
        monkeypatch.setenv("SA_FORCE_GROUP", variant)
    prob = make_problem("robertson")
    d = robertson_batch(9)
    tv = d["tvals"]
    atol = np.array([1e-8, 1e-13, 1e-7])
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(3)[None, :])
    sol = AdjointSolver(prob, abstol=atol, reltol=1e-7, backward_abstol=1e-9, backward_reltol=1e-7,
                        quad_abstol=1e-8, quad_reltol=1e-7)
    y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], d["params"], np.zeros(0))
    g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    orc = make_oracle("robertson")
    cfg = orc.config(rtol=1e-7, atol=atol, rtolB=1e-7, atolB=1e-9, rtolQB=1e-7, atolQB=1e-8)
    yo, so, sto = orc.solve_forward(cfg, d["y0"], d["params"], np.zeros(0), 0.0, tv)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads)
    assert (st == 0).all() and (stb == 0).all()
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)
    plain = Solver(prob, abstol=atol, reltol=1e-7)
    yp, stp, statsp = plain.solve_batch(0.0, tv, d["y0"], d["params"], np.zeros(0))
    ypo, spo, stpo = orc.solve(cfg, d["y0"], d["params"], np.zeros(0), 0.0, tv)
    np.testing.assert_array_equal(yp, ypo)
    np.testing.assert_array_equal(statsp[:, CMP[:8]], stpo[:, CMP[:8]])


def _oracle_adjoint(name, cfg_kw, y0, ps, pr, t0, tv, grads, t_start=None, t_end=None):
    orc = make_oracle(name)
    cfg = orc.config(**cfg_kw)
    y, st, stats = orc.solve_forward(cfg, y0, ps, pr, t0, tv)
    g, lam, stb, statsb = orc.solve_backward(cfg, tv[-1] if t_start is None else t_start,
                                             t0 if t_end is None else t_end, tv, grads)
    return y, st, g, lam, stb


@pytest.mark.parametrize("variant", [None, "8", "wave4", "wave", "mem"])
def test_edge_cases_match_oracle(variant, monkeypatch):
    """Ragged / degenerate inputs (reference semantics, solver.py:705-708, 750-776):
    tvals[0] > t0 (extra interval down to tend), a single output time, B = 1, B not a multiple of 64,
    per-instance cotangents, tvals containing t0 twice -- in every mapping."""
    from sunode_amd.solver import AdjointSolver
    if variant:
        monkeypatch.setenv("SA_FORCE_GROUP", variant)
    prob = make_problem("lv")
    tol = dict(abstol=1e-8, reltol=1e-8, backward_abstol=1e-8, backward_reltol=1e-8, quad_abstol=1e-8, quad_reltol=1e-8)
    okw = dict(rtol=1e-8, atol=1e-8, rtolB=1e-8, atolB=1e-8, rtolQB=1e-8, atolQB=1e-8)
    rng = np.random.RandomState(5)
    for B, tv, t0 in [(1, np.linspace(0, 10), 0.0), (67, np.array([2.5]), 0.0), (3, np.linspace(1.0, 4.0, 7), 0.5),
                      (5, np.array([0.0, 0.0, 1.0, 3.0]), 0.0)]:
        d = lv_batch(B)
        ps, pr = d["params"][:, :2], d["params"][:, 2:]
        grads = rng.randn(B, len(tv), 2)
        sol = AdjointSolver(prob, **tol)
        y, st, _ = sol.solve_forward_batch(t0, tv, d["y0"], ps, pr)
        g, lam, stb, _ = sol.solve_backward_batch(tv[-1], t0, tv, grads)
        yo, so, go, lo, sbo = _oracle_adjoint("lv", okw, d["y0"], ps, pr, t0, tv, grads)
        np.testing.assert_array_equal(st, so)
        np.testing.assert_array_equal(stb, sbo)
        np.testing.assert_array_equal(y, yo)
        np.testing.assert_array_equal(g, go)
        np.testing.assert_array_equal(lam, lo)


def test_empty_batch_and_no_derivative_params():
    from sunode_amd import SympyProblem
    from sunode_amd.solver import AdjointSolver, Solver
    prob = make_problem("lv")
    sol = Solver(prob, abstol=1e-8, reltol=1e-8)
    y, st, stats = sol.solve_batch(0.0, np.linspace(0, 1, 5), np.zeros((0, 2)), np.zeros((0, 2)), np.zeros((0, 2)))
    assert y.shape == (0, 5, 2) and st.shape == (0,)
    # a problem without differentiated parameters: quadrature length 0 (reference test_solve.py:7-78)
    p0 = SympyProblem({"b": ()}, {"x": ()}, lambda t, y, p: {"x": -p.b * y.x}, derivative_params=[])
    adj = AdjointSolver(p0, abstol=1e-10, reltol=1e-10)
    tv = np.linspace(0, 1, 11)
    y, st, _ = adj.solve_forward_batch(0.0, tv, np.ones((4, 1)), np.zeros((4, 0)), np.array([2.0]))
    g, lam, stb, _ = adj.solve_backward_batch(tv[-1], 0.0, tv, np.ones((11, 1)))
    assert (st == 0).all() and (stb == 0).all() and g.shape == (4, 0)
    np.testing.assert_allclose(y[0, :, 0], np.exp(-2.0 * tv), rtol=2e-8)
    np.testing.assert_allclose(-lam[:, 0], np.exp(-2.0 * tv).sum(), rtol=1e-7)      # dL/dy0, L = sum_k y(t_k)


def test_backward_reports_failed_forward_and_arena_limit():
    """Instances that would store more than max_steps points come back SA_STATUS_ARENA_FULL (-9001, NOT
    CV_TOO_MUCH_WORK) from the forward call -- the integration stops at the bound (no unbounded pass), outputs NaN
    like every other failure -- and CV_NO_FWD (-102) with NaN gradients from the backward call; the other instances
    of the batch are unaffected."""
    from sunode_amd.solver import AdjointSolver
    prob = make_problem("robertson")
    d = robertson_batch(8)
    tv = d["tvals"]
    params = d["params"].copy()
    kw = dict(abstol=1e-10, reltol=1e-8, backward_abstol=1e-10, backward_reltol=1e-8, quad_abstol=1e-10,
              quad_reltol=1e-8)
    ref = AdjointSolver(prob, **kw)
    yr, sr, statr = ref.solve_forward_batch(0.0, tv, d["y0"], params, np.zeros(0))
    gr, lr, sbr, _ = ref.solve_backward_batch(tv[-1], 0.0, tv, np.ones((len(tv), 3)))
    assert (sr == 0).all() and (sbr == 0).all()
    cap = int(np.sort(statr[:, 8])[3])                      # the four shortest trajectories fit
    sol = AdjointSolver(prob, max_steps=cap, **kw)
    y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], params, np.zeros(0))
    g, lam, stb, _ = sol.solve_backward_batch(tv[-1], 0.0, tv, np.ones((len(tv), 3)))
    full = statr[:, 8] > cap
    assert full.any() and not full.all()
    assert (st[full] == -9001).all() and (st[~full] == 0).all()
    assert ((stb == -102) == full).all() and (stb[~full] == 0).all()
    np.testing.assert_array_equal(y[~full], yr[~full])
    assert np.isnan(y[full]).all() and (stats[full, 8] == cap).all()       # stopped at the bound
    assert np.isnan(g[full]).all() and np.isnan(lam[full]).all()
    np.testing.assert_array_equal(g[~full], gr[~full])
    np.testing.assert_array_equal(lam[~full], lr[~full])


@pytest.mark.gpu
@pytest.mark.parametrize("group", [None, "8", "wave16", "wave32"])
def test_seir_device_counters_equal_dvode(group, golden_dir, monkeypatch):
    """an independent counter oracle for the mid-size mappings.  SEIR (n = 16) forward through the
    default lane-group kernel (4 lanes per instance, LU factors in registers), the lean build with 8 lanes, 16 lanes
    and 32 lanes (matrix in LDS): every step counter equals Fortran DVODE's
    (tests/golden/dvode_seir.json), states to round-off."""
    import json
    from sunode_amd.solver import AdjointSolver
    with open(os.path.join(golden_dir, "dvode_seir.json")) as fh:
        dv = json.load(fh)
    if group:
        monkeypatch.setenv("SA_FORCE_GROUP", group)
    prob = make_problem("seir")
    cases = [dv["seir_batch_%d" % b] for b in range(4)]
    tv = np.array(cases[0]["tvals"])
    sol = AdjointSolver(prob, abstol=1e-8, reltol=1e-8, backward_abstol=1e-8, backward_reltol=1e-8,
                        quad_abstol=1e-8, quad_reltol=1e-8)
    y, st, stats = sol.solve_forward_batch(0.0, tv, np.array([c["y0"] for c in cases]),
                                           np.array([c["ps"] for c in cases]), np.array(cases[0]["pr"]))
    assert (st == 0).all()
    for b, c in enumerate(cases):
        got = [int(v) for v in stats[b][:8]]
        assert got == [c["nst"], c["nfe"], c["nlu"], c["nje"], c["nni"], c["ncfn"], c["netf"], c["qlast"]]
        assert stats[b][8] == c["nst"] + 1                      # stored data points
        ref = np.array(c["y"])
        np.testing.assert_allclose(y[b], ref, rtol=0, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("group", [None, "wave4"])
def test_device_backward_controller_equals_dvode(group, golden_dir, monkeypatch):
    """The adjoint pass ON THE DEVICE against Fortran DVODE (tests/golden/dvode_backward.json, see
    tests/test_oracle_pinning.py::test_backward_controller_equals_dvode): one interval T -> t_mid per call
    (solve_backward_batch(t0=T, tend=t_mid)), quadrature tolerances so loose that the built-in quadrature error
    control never decides anything; every counter equal on the exact rows, lambda(t_mid) to round-off level.
    All cases of a problem go through one batch per t_mid index."""
    import json
    from sunode_amd.solver import AdjointSolver
    if group:
        monkeypatch.setenv("SA_FORCE_GROUP", group)
    with open(os.path.join(golden_dir, "dvode_backward.json")) as fh:
        gold = json.load(fh)
    exact = lambda tag, t_mid: not (tag.startswith("robertson") and not (tag == "robertson_0" and t_mid == 30.0))  # noqa: E731
    n_exact = 0
    for tag, c in gold.items():
        if group and c["problem"] != "seir":
            continue
        prob = make_problem(c["problem"])
        sol = AdjointSolver(prob, abstol=c["atol"], reltol=c["rtol"], backward_abstol=c["atol"], backward_reltol=c["rtol"],
                            quad_abstol=1e30, quad_reltol=0.0)
        tv = np.array([c["T"]])
        y, st, sf = sol.solve_forward_batch(0.0, tv, np.array([c["y0"]]), np.array([c["ps"]]), np.array(c["pr"]))
        assert st[0] == 0 and sf[0][8] == len(c["fwd_t"])
        for row in c["intervals"]:
            g, lam, stb, sb = sol.solve_backward_batch(c["T"], row["t_mid"], tv, np.array(c["g"])[None, None, :])
            assert stb[0] == 0
            got = [int(v) for v in sb[0][:8]]
            want = [row[k] for k in ("nst", "nfe", "nlu", "nje", "nni", "ncfn", "netf", "qlast")]
            ref = np.array(row["lam_mid"])
            if exact(tag, row["t_mid"]):
                assert got == want, (tag, row["t_mid"], got, want)
                np.testing.assert_allclose(lam[0], ref, rtol=0, atol=(1e-8 if tag == "robertson_0" else 5e-12) * np.abs(ref).max())
                n_exact += 1
    assert n_exact == (6 if group else 23)


@pytest.mark.parametrize("name,variant", [("network24", None), ("network24", "wave32"), ("network24", "wave"),
                                          ("network24", "mem"), ("network100", None)])
def test_device_structured_callbacks_match_reference(name, variant, golden_dir, monkeypatch):
    """The structured (SA_MATVEC / SA_MATFILL / SA_SUM / SA_ROLLED) callbacks ON THE DEVICE, lane-parallel in the
    lane-group / workgroup mappings, against the reference's own lambdify output
    (tests/golden/callbacks_network.json) -- and bit-equal to the host build of the same generated source."""
    from sunode_amd.solver import Solver
    from tests.helpers import check_matrix_summary, network_golden_points
    if variant:
        monkeypatch.setenv("SA_FORCE_GROUP", variant)
    prob = make_problem(name)
    n, pts = network_golden_points(golden_dir, name)
    eng = Solver(prob)._engine()
    pr = np.array([prob.extend_remainder(pt["K"]) for pt in pts])
    got = eng.eval_callbacks([pt["t"] for pt in pts], [pt["y"] for pt in pts], [pt["lam"] for pt in pts],
                             np.array([pt["scale"] for pt in pts]), pr)
    orc = make_oracle(name)
    for i, pt in enumerate(pts):
        host = orc.eval(pt["t"], pt["y"], pt["lam"], np.array(pt["scale"]), pt["K"])
        for key in ("rhs", "adj", "quad"):
            ref = np.array(pt[key])
            np.testing.assert_allclose(got[key][i], ref, rtol=1e-12, atol=64 * 2.3e-16 * np.abs(ref).max(), err_msg=key)
            np.testing.assert_array_equal(got[key][i], host[key])
        for key in ("jac", "adjjac"):
            check_matrix_summary(got[key][i], pt[key])
            np.testing.assert_array_equal(got[key][i], host[key])
        assert got["codes"][i].tolist() == pt["codes"]


@pytest.mark.parametrize("name", ["lv", "robertson", "seir", "network24"])
def test_compact_trajectory_equals_table_records(name):
    """AdjointSolver(compact_trajectory=True): {order, t, y[n]} per stored step, divided-difference table rebuilt in the
    backward kernel on every index move -- everything (states, gradients, all counters incl. the number of rebuilds)
    bit-identical to the default table records, in a fifth of the arena.  lv / robertson: one lane per instance;
    seir: lean lane groups (table in LDS); network24: lane groups with the table in registers."""
    from sunode_amd.solver import AdjointSolver
    prob = make_problem(name)
    B = 300
    if name == "lv":
        d = lv_batch(B); ps, pr = d["params"][:, :2], d["params"][:, 2:]; rt, at = 1e-8, 1e-8
    elif name == "seir":
        B = 70
        d = seir_batch(B); ps, pr = d["ps"], d["pr"]; rt, at = 1e-8, 1e-8
    elif name == "network24":
        B = 20
        d = network_batch(B, 24); ps, pr = d["ps"], d["pr"]; rt, at = d["rtol"], d["atol"]
    else:
        d = robertson_batch(B); ps, pr = d["params"], np.zeros(0); rt, at = 1e-8, 1e-10
    tv = d["tvals"]
    n = prob.n_states
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(n)[None, :])
    res = []
    for compact in (False, True):
        sol = AdjointSolver(prob, abstol=at, reltol=rt, backward_abstol=at, backward_reltol=rt, quad_abstol=at,
                            quad_reltol=rt, compact_trajectory=compact)
        y, st, sf = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
        g, lam, stb, sb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
        assert (st == 0).all() and (stb == 0).all()
        res.append((y, sf[:, CMP], g, lam, sb[:, CMP_B], sol._engine().arena_info()[0]))
        sol._engine().close()
    for a, b in zip(res[0][:5], res[1][:5]):
        np.testing.assert_array_equal(a, b)
    assert res[1][5] * 4 < res[0][5]            # arena bytes: (n + 2) against (8 + 6 n) doubles per point


@pytest.mark.parametrize("name,compact", [("robertson", None), ("lv", True)])
def test_index_search_equals_cvodes_walk(name, compact, monkeypatch):
    """Round 6: the compact-record builds of the one-lane mapping find the interpolation index by table times + the
    remembered right neighbour + galloping / section search (bdf_kernels.hip SA_SEARCH_CACHE, search_left / search_right)
    instead of CVAfindIndex' walk over the stored points.  Same index, same bracket times: states, gradients and EVERY
    counter (interpolations and table rebuilds included) equal the -DSA_SEARCH_CACHE=0 build, which walks like CVODES and
    the oracle -- on Robertson (801 + 258 points walked per instance) and on LV with compact records (table in
    registers).  The algorithm itself: tests/test_index_search.py."""
    from sunode_amd.solver import AdjointSolver
    prob = make_problem(name)
    B = 320
    if name == "lv":
        d = lv_batch(B); ps, pr = d["params"][:, :2], d["params"][:, 2:]; rt, at = 1e-8, 1e-8
    else:
        d = robertson_batch(B); ps, pr = d["params"], np.zeros(0); rt, at = 1e-8, 1e-10
    tv = d["tvals"]
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(prob.n_states)[None, :])
    res = []
    for defines in ("", "-DSA_SEARCH_CACHE=0"):
        if defines:
            monkeypatch.setenv("SA_KERNEL_DEFINES", defines)
        sol = AdjointSolver(prob, abstol=at, reltol=rt, backward_abstol=at, backward_reltol=rt, quad_abstol=at,
                            quad_reltol=rt, compact_trajectory=compact)
        y, st, sf = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
        g, lam, stb, sb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
        assert (st == 0).all() and (stb == 0).all()
        res.append((y, sf[:, :14], g, lam, sb[:, :14]))
        sol._engine().close()
    for a, b in zip(res[0], res[1]):
        np.testing.assert_array_equal(a, b)
    assert res[0][4][:, 12].min() > 0              # (the table was rebuilt: the index did move)


@pytest.mark.parametrize("defines", ["-DSA_SEARCH_COUNT", "-DSA_ABLATE_PROFILE -DSA_INTERP_PROFILE"])
def test_diagnostic_builds_of_the_one_lane_kernel_integrate_like_the_default_build(defines, monkeypatch):
    """tools/profile_lv.py's builds of bdf_kernels.hip (the numbers in profiles/r06_interp_search.txt and *_lv_phases.txt
    come from them): -DSA_SEARCH_COUNT counts the index search's dependent loads in the statistics slots of the
    interpolations / rebuilds, -DSA_ABLATE_PROFILE -DSA_INTERP_PROFILE puts clock readings into slots 8..15.  Neither may
    change a result: states, gradients, adjoint states and the CVODES counters equal the default build's."""
    from sunode_amd.solver import AdjointSolver
    prob = make_problem("robertson")
    B = 96
    d = robertson_batch(B)
    tv = d["tvals"]
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(3)[None, :])
    res = []
    for env in ("", defines):
        if env:
            monkeypatch.setenv("SA_KERNEL_DEFINES", env)
        sol = AdjointSolver(prob, abstol=1e-10, reltol=1e-8, backward_abstol=1e-10, backward_reltol=1e-8,
                            quad_abstol=1e-10, quad_reltol=1e-8)
        y, st, sf = sol.solve_forward_batch(0.0, tv, d["y0"], d["params"], np.zeros(0))
        g, lam, stb, sb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
        assert (st == 0).all() and (stb == 0).all()
        res.append((y, g, lam, sf[:, :8], sb[:, :8], sb))
        sol._engine().close()
    for a, b in zip(res[0][:5], res[1][:5]):
        np.testing.assert_array_equal(a, b)
    if "COUNT" in defines:
        assert res[1][5][:, 11].sum() > 0 and res[1][5][:, 12].sum() > 0      # far moves to the left and to the right happened
    else:
        assert res[1][5][:, 9].min() > 0                                       # the search's timer did run


def test_stiff_five_state_model_equals_oracle():
    """robertson5 (tools/problems.py): a stiff model with five states and four quadratures in the one-lane mapping, compact
    records -- 1 260 backward steps over 720 stored points, the index moves in a third of the attempts and then by
    several points (the far path of the index search at another vector length than Robertson's).  States, gradients,
    adjoint states and every counter equal the oracle's, which walks like CVAfindIndex."""
    from sunode_amd.solver import AdjointSolver
    from tools.problems import robertson5_batch
    prob = make_problem("robertson5")
    B = 200
    d = robertson5_batch(B)
    tv = d["tvals"]
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(5)[None, :])
    sol = AdjointSolver(prob, abstol=1e-10, reltol=1e-8, backward_abstol=1e-10, backward_reltol=1e-8,
                        quad_abstol=1e-10, quad_reltol=1e-8)
    y, st, sf = sol.solve_forward_batch(0.0, tv, d["y0"], d["params"], np.zeros(0))
    g, lam, stb, sb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    orc = make_oracle("robertson5")
    cfg = orc.config(rtol=1e-8, atol=1e-10, rtolB=1e-8, atolB=1e-10, rtolQB=1e-8, atolQB=1e-10)
    yo, so, sto = orc.solve_forward(cfg, d["y0"], d["params"], np.zeros(0), 0.0, tv, nthreads=8)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=8)
    assert (st == 0).all() and (stb == 0).all()
    np.testing.assert_array_equal(st, so); np.testing.assert_array_equal(stb, sbo)
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)
    np.testing.assert_array_equal(sf[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(sb[:, CMP_B], stbo[:, CMP_B])
    assert sb[:, 12].mean() * 2 < sb[:, 0].mean()          # fewer table rebuilds than steps: the index jumps


@pytest.mark.parametrize("name", ["lv", "robertson", "seir"])
def test_randomized_sweep_matches_oracle(name):
    """Draws far outside the BASELINE batches: parameters spread over an order of magnitude (including draws whose
    solves fail or hit the step budget), three tolerance settings with DIFFERENT forward / backward / quadrature
    tolerances, irregular output grids with repeated and t0-valued entries, per-instance cotangents -- states,
    gradients, adjoint states, every status code and every counter must equal the oracle's, instance by instance."""
    from sunode_amd.solver import AdjointSolver
    prob = make_problem(name)
    rng = np.random.RandomState({"lv": 11, "robertson": 12, "seir": 13}[name])
    B = {"lv": 2048, "robertson": 1024, "seir": 192}[name]
    if name == "lv":
        d = lv_batch(B); p0 = d["params"]; y0 = d["y0"] * np.exp(0.5 * rng.randn(B, 2)); T = 10.0
        par = p0 * np.exp(0.8 * rng.randn(B, 4)); ps, pr = par[:, :2], par[:, 2:]
    elif name == "robertson":
        d = robertson_batch(B); y0 = d["y0"]; T = 400.0
        ps, pr = d["params"] * np.exp(0.7 * rng.randn(B, 3)), np.zeros(0)
    else:
        d = seir_batch(B); y0 = d["y0"]; T = 60.0
        ps, pr = d["ps"] * np.exp(0.6 * rng.randn(B, 8)), d["pr"]
    n = prob.n_states
    orc = make_oracle(name)
    for k, (rt, at, rtb, atb, rtq, atq) in enumerate([(1e-6, 1e-8, 1e-5, 1e-7, 1e-4, 1e-6),
                                                     (1e-9, 1e-11, 1e-8, 1e-9, 1e-8, 1e-8),
                                                     (1e-4, 1e-6, 1e-6, 1e-9, 1e-7, 1e-9)]):
        inner = np.sort(rng.uniform(0.0, T, 9))
        tv = np.concatenate([[0.0], inner[:4], inner[3:4], inner[4:], [T]])       # starts at t0, one repeated time
        grads = rng.randn(B, len(tv), n)
        sol = AdjointSolver(prob, abstol=at, reltol=rt, backward_abstol=atb, backward_reltol=rtb, quad_abstol=atq,
                            quad_reltol=rtq, mxsteps=400)
        y, st, sf = sol.solve_forward_batch(0.0, tv, y0, ps, pr)
        g, lam, stb, sb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
        cfg = orc.config(rtol=rt, atol=at, rtolB=rtb, atolB=atb, rtolQB=rtq, atolQB=atq, mxstep=400)
        yo, so, sfo = orc.solve_forward(cfg, y0, ps, pr, 0.0, tv, nthreads=os.cpu_count() or 8)
        go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=os.cpu_count() or 8)
        np.testing.assert_array_equal(st, so, err_msg="forward status, setting %d" % k)
        np.testing.assert_array_equal(stb, sbo, err_msg="backward status, setting %d" % k)
        ok = (so == 0)
        np.testing.assert_array_equal(sf[ok][:, CMP], sfo[ok][:, CMP])
        np.testing.assert_array_equal(y[ok], yo[ok])
        okb = ok & (sbo == 0)
        assert okb.sum() > B // 2, "too few successful draws for a meaningful comparison: %d" % okb.sum()
        np.testing.assert_array_equal(sb[okb][:, CMP_B], stbo[okb][:, CMP_B])
        np.testing.assert_array_equal(g[okb], go[okb])
        np.testing.assert_array_equal(lam[okb], lo[okb])
        assert np.isnan(y[~ok]).all() and np.isnan(g[~okb]).all()
        sol._engine().close()


@pytest.mark.parametrize("n,group", [(100, None), (100, "mem"), (24, None), (24, "wave"), (24, "mem")])
def test_network_device_counters_equal_dvode(n, group, golden_dir, monkeypatch):
    """a counter oracle that is not ours at config 5's size.  The 100-state network forward through
    the workgroup-per-instance kernel (panel LU in registers, matrix-vector callbacks) and the memory-resident one,
    the 24-state network through 16-lane groups, a forced workgroup and the memory-resident kernel: every step counter
    equals Fortran DVODE's (tests/golden/dvode_network.json: numpy restatement of the model), states to round-off."""
    import json
    from sunode_amd.solver import AdjointSolver
    with open(os.path.join(golden_dir, "dvode_network.json")) as fh:
        dv = json.load(fh)
    if group:
        monkeypatch.setenv("SA_FORCE_GROUP", group)
    prob = make_problem("network%d" % n)
    cases = [dv["network%d_batch_%d" % (n, b)] for b in range(4)]
    d = network_batch(4, n=n)
    assert d["ps"].tolist() == [c["ps"] for c in cases]
    tv = np.array(cases[0]["tvals"])
    sol = AdjointSolver(prob, abstol=1e-8, reltol=1e-8, backward_abstol=1e-8, backward_reltol=1e-8,
                        quad_abstol=1e-8, quad_reltol=1e-8, max_steps=1024)
    y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], d["ps"], d["pr"])
    assert (st == 0).all()
    for b, c in enumerate(cases):
        got = [int(v) for v in stats[b][:8]]
        assert got == [c["nst"], c["nfe"], c["nlu"], c["nje"], c["nni"], c["ncfn"], c["netf"], c["qlast"]]
        assert stats[b][8] == c["nst"] + 1                      # stored data points
        ref = np.array(c["y"])
        np.testing.assert_allclose(y[b], ref, rtol=0, atol=1e-11 * np.abs(ref).max())


@pytest.mark.parametrize("n,group", [(100, None), (24, None), (24, "wave"), (24, "mem")])
def test_network_device_gradients_match_truth(n, group, golden_dir, monkeypatch):
    """dL/dp and dL/dy0 of the networks from the DEVICE against DOP853 on the sensitivity equations
    (tests/golden/truth_network*.npz, non-trivial cotangent): 2e-6 of the largest component at rtol = atol = 1e-8."""
    from sunode_amd.solver import AdjointSolver
    if group:
        monkeypatch.setenv("SA_FORCE_GROUP", group)
    d = np.load(os.path.join(golden_dir, "truth_network%d.npz" % n))
    B = len(d["ps"])
    pr = network_batch(B, n=n)["pr"]
    prob = make_problem("network%d" % n)
    tol = 1e-8
    sol = AdjointSolver(prob, abstol=tol, reltol=tol, backward_abstol=tol, backward_reltol=tol,
                        quad_abstol=tol, quad_reltol=tol, max_steps=1024)
    tv = d["tvals"]
    y, st, _ = sol.solve_forward_batch(0.0, tv, d["y0"], d["ps"], pr)
    g, lam, st2, _ = sol.solve_backward_batch(tv[-1], 0.0, tv, d["grads"])
    assert (st == 0).all() and (st2 == 0).all()
    assert np.max(np.abs(y - d["y_out"]) / np.abs(d["y_out"]).max(axis=(0, 1))) < 1e-6
    gt = d["grad_params"]
    assert np.max(np.abs(g - gt) / np.abs(gt).max(axis=1, keepdims=True)) < 2e-6
    assert np.max(np.abs(-lam - d["grad_y0"]) / np.abs(d["grad_y0"]).max(axis=1, keepdims=True)) < 2e-6


@pytest.mark.parametrize("name", ["lv", "seir"])
def test_conservative_build_without_the_vgpr_liverange_pass(name, monkeypatch):
    """SA_VGPR_LIVERANGE_OPT=0: every register-resident kernel built with -amdgpu-opt-vgpr-liverange=0 (round 4 traced a
    miscompile of an unshipped variant of the 4-lane sensitivity build to SIOptimizeVGPRLiveRange:
    profiles/r04_sens_anomaly.txt).  The conservative build is another code object (own cache key) with the same
    results: forward + adjoint of LV (one lane per instance) and SEIR (4-lane groups) bit-equal to the oracle, and SEIR's
    forward sensitivities too."""
    from sunode_amd import _native
    from sunode_amd.solver import AdjointSolver, Solver
    prob = make_problem(name)
    fast = _native.code_object_path(prob.native_source(), compact=_native.default_compact_trajectory(prob.native_source()))
    monkeypatch.setenv("SA_VGPR_LIVERANGE_OPT", "0")
    assert _native._safety_flags() == _native.SAFETY_CODEGEN_FLAGS.split()
    assert _native.code_object_path(prob.native_source(), compact=_native.default_compact_trajectory(prob.native_source())) != fast
    B = 130
    if name == "lv":
        d = lv_batch(B); ps, pr = d["params"][:, :2], d["params"][:, 2:]
    else:
        d = seir_batch(B); ps, pr = d["ps"], d["pr"]
    tv = d["tvals"]
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(prob.n_states)[None, :])
    tol = 1e-8
    sol = AdjointSolver(prob, abstol=tol, reltol=tol, backward_abstol=tol, backward_reltol=tol, quad_abstol=tol, quad_reltol=tol)
    y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
    g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    orc = make_oracle(name)
    cfg = orc.config(rtol=tol, atol=tol, rtolB=tol, atolB=tol, rtolQB=tol, atolQB=tol)
    yo, so, sto = orc.solve_forward(cfg, d["y0"], ps, pr, 0.0, tv, nthreads=8)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=8)
    assert (st == 0).all() and (stb == 0).all()
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(statsb[:, CMP_B], stbo[:, CMP_B])
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)
    if name == "seir":
        sens0 = np.zeros((prob.n_params, prob.n_states))
        s2 = Solver(prob, abstol=tol, reltol=tol, sens_mode="simultaneous")
        ys, S, sts, _ = s2.solve_sens_batch(0.0, tv[::5], d["y0"][:21], ps[:21], pr, sens0)
        yso, So, _, _ = orc.solve_sens(orc.config(rtol=tol, atol=tol), d["y0"][:21], ps[:21], pr, sens0, 0.0, tv[::5],
                                       mode="simultaneous", nthreads=8)
        np.testing.assert_array_equal(ys, yso)
        np.testing.assert_array_equal(S, So)



@pytest.mark.parametrize("name,mapping,defines", [
    ("seir", None, "-DSA_WAVE_PROFILE"),
    ("seir", None, "-DSA_WAVE_PROFILE -DSA_WAVE_PROFILE_PHASES"),
    ("network24", "wave", "-DSA_WAVE_PROFILE"),
    ("network24", "wave", "-DSA_WAVE_PROFILE -DSA_LU_PROFILE_SEGMENTS"),
    ("network24", "wave", "-DSA_WAVE_PROFILE -DSA_LU_PROFILE_TIMELINE"),
    ("network24", "wave", "-DSA_LU_INLINE_WORKERS"),          # (measured, slower: see worker_loop)
])
def test_profiling_builds_integrate_like_the_default_build(name, mapping, defines, monkeypatch):
    """The section-timer builds of bdf_wave.hip (tools/profile_wave.py; the numbers under profiles/*_sections.txt and
    *_lu.txt come from them) put clock readings into the upper statistics slots and must change nothing else: states,
    gradients and the CVODES counters equal the oracle's.  (Also: every conditional of the kernel file is reached by a
    default build or by a GPU test.)"""
    from sunode_amd.solver import AdjointSolver
    monkeypatch.setenv("SA_KERNEL_DEFINES", defines)
    if mapping:
        monkeypatch.setenv("SA_FORCE_GROUP", mapping)
    prob = make_problem(name)
    B = 9
    d = seir_batch(B) if name == "seir" else network_batch(B, n=24)
    tv = d["tvals"][::10] if name == "seir" else d["tvals"][::2]
    k = np.arange(len(tv))[:, None]; i = np.arange(prob.n_states)[None, :]
    grads = 1.0 + 0.5 * np.cos(1.7 * k + 0.9 * i)
    sol = AdjointSolver(prob, abstol=1e-8, reltol=1e-8, backward_abstol=1e-8, backward_reltol=1e-8,
                        quad_abstol=1e-8, quad_reltol=1e-8)
    y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], d["ps"], d["pr"])
    g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    orc = make_oracle(name)
    cfg = orc.config(rtol=1e-8, atol=1e-8, rtolB=1e-8, atolB=1e-8, rtolQB=1e-8, atolQB=1e-8)
    yo, so, sto = orc.solve_forward(cfg, d["y0"], d["ps"], d["pr"], 0.0, tv, nthreads=8)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=8)
    assert (st == 0).all() and (stb == 0).all()
    if "LU_PROFILE" not in defines:                                  # (the LU timers take more of the slots)
        np.testing.assert_array_equal(stats[:, :5], sto[:, :5])      # (slots 5.. carry clock readings in these builds)
        np.testing.assert_array_equal(statsb[:, :5], stbo[:, :5])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)
    assert "PROFILE" not in defines or statsb[:, 15].min() > 0        # the timers did run


@pytest.mark.parametrize("mapping", [None, "4", "mem"])
def test_lane_family_callbacks_in_the_one_lane_and_other_mappings(mapping, monkeypatch):
    """Two age groups x (S, I): the generated callbacks are lane families of two members (symode/codegen.py
    find_lane_families).  The one-lane-per-instance kernel unrolls the member loop at compile time (register arrays),
    4 lanes per instance give lanes 0..1 / 2..3 the members 0 / 1, the memory-resident mapping loops: callbacks equal
    the host build bit for bit, and the whole forward + adjoint solve equals the oracle."""
    from sunode_amd import _native
    from sunode_amd.solver import AdjointSolver
    if mapping:
        monkeypatch.setenv("SA_FORCE_GROUP", mapping)
    prob = make_problem("sir2")
    assert _native.kernel_variant(prob.native_source())[0] == {None: "bdf_kernels.hip", "4": "bdf_wave.hip", "mem": "bdf_mem.hip"}[mapping]
    rng = np.random.RandomState(11)
    B = 37
    y0 = np.tile([900.0, 700.0, 3.0, 1.0], (B, 1)) * np.exp(0.05 * rng.randn(B, 4))
    ps = np.array([0.4, 0.3, 0.15]) * np.exp(0.2 * rng.randn(B, 3))
    pr = np.array([1.0, 0.3, 0.2, 1.0, 50.0, 30.0])
    tv = np.linspace(0.0, 40.0, 9)
    grads = 1.0 + 0.5 * np.cos(1.1 * np.arange(9)[:, None] + 0.7 * np.arange(4)[None, :])
    tol = dict(abstol=1e-8, reltol=1e-8, backward_abstol=1e-8, backward_reltol=1e-8, quad_abstol=1e-8, quad_reltol=1e-8)
    sol = AdjointSolver(prob, batch_mapping="fixed", **tol)     # (the mapping under test, not the small-batch switch)
    orc = make_oracle("sir2")
    tpts, lam5 = np.linspace(0.0, 1.0, 5), rng.randn(5, 4)
    got = sol._engine().eval_callbacks(tpts, y0[:5], lam5, ps[:5], np.tile(pr, (5, 1)))
    for i in range(5):
        host = orc.eval(tpts[i], y0[i], lam5[i], ps[i], pr)
        for key in ("rhs", "adj", "quad"):
            np.testing.assert_array_equal(got[key][i], host[key])
    y, st, stats = sol.solve_forward_batch(0.0, tv, y0, ps, pr)
    g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
    cfg = orc.config(rtol=1e-8, atol=1e-8, rtolB=1e-8, atolB=1e-8, rtolQB=1e-8, atolQB=1e-8)
    yo, so, sto = orc.solve_forward(cfg, y0, ps, pr, 0.0, tv, nthreads=8)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=8)
    assert (st == 0).all() and (stb == 0).all() and (so == 0).all() and (sbo == 0).all()
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(statsb[:, CMP_B], stbo[:, CMP_B])
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)
    assert np.abs(g).max() > 0
