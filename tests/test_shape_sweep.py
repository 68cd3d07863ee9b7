"""Parity over the model-shape space: the DEFAULT build of every (n, p) band and boundary vs the CPU oracle.

The product is a code generator: every user model is a new code object whose kernel family follows from (n, p)
(sunode_amd/_native.py ``kernel_variant``).  The BASELINE problems cover n in {2, 3, 16, 100}; this sweep runs a
generated model family (tools/problems.py ``random_network`` -- the reference takes any sympy system,
/root/reference/sunode/symode/problem.py:25-33) through the mapping the engine selects BY ITSELF (no SA_FORCE_GROUP)
on both sides of every selection boundary.  Bar: statuses, every counter and every fp64 output ``array_equal``.
"""
import os

import numpy as np
import pytest

from tests.helpers import make_oracle, make_problem
from tools.sweep_cases import ADJOINT_CASES, SENS_CASES, batch_of

CMP = [0, 1, 2, 3, 4, 5, 6, 7, 8]
CMP_B = [0, 1, 2, 3, 4, 5, 6, 9, 10, 12]


def test_sweep_covers_every_mapping_family():
    """CPU: the shapes of the sweep reach every kernel family and both sides of every boundary of the selection."""
    from sunode_amd import _native
    seen = set()
    for name, _ in ADJOINT_CASES:
        assert not os.environ.get("SA_FORCE_GROUP")
        prob = make_problem(name) if name in ("lv12", "rn5_8", "rn6_1") else None
        import re
        m = re.fullmatch(r"rnb?(\d+)_(\d+)", name)
        c = re.fullmatch(r"chain(\d+)", name)
        n, p = (2, 12) if name == "lv12" else ((int(c.group(1)), 2) if c else (int(m.group(1)), int(m.group(2))))
        src = "#define SA_N_STATES %d\n#define SA_N_SUB %d\n" % (n, p)
        fam = _native.kernel_variant(src)
        if prob is not None:
            assert fam == _native.kernel_variant(prob.native_source())
        lean = fam[0] == "bdf_wave.hip" and fam[1] <= 8 and n * ((n + fam[1] - 1) // fam[1]) <= 64
        seen.add((fam[0], fam[1], lean))
    assert {("bdf_kernels.hip", 1, False), ("bdf_wave.hip", 4, True), ("bdf_wave.hip", 8, True),
            ("bdf_wave.hip", 16, False), ("bdf_wave.hip", 32, False), ("bdf_wave.hip", 64, False),
            ("bdf_mem.hip", 1, False)} <= seen


# ---- an independent pin for the sweep's callbacks: the reference's own pipeline, one shape per kernel family ----
PINNED = ["lv12", "rn7_4", "rnb15_9", "rn22_33", "rn65_4"]


def _pinned_points(name, golden_dir):
    """(fixture, t, y, lam, ps, pr) of tests/golden/callbacks_sweep.json (tools/make_golden_callbacks_sweep.py:
    values from the reference's symbolic pipeline; inputs regenerated from the repo-owned streams)."""
    import json
    from tools.make_golden_callbacks_sweep import sweep_points
    with open(os.path.join(golden_dir, "callbacks_sweep.json")) as fh:
        fix = json.load(fh)[name]
    prob = make_problem(name)
    t, y, lam, par = sweep_points(name, fix["n"], fix["n_items"])
    return fix, prob, t, y, lam, par[:, prob.params_subset.subset_index], par[:, prob.params_subset.remainder_index]


def _check_against_reference(fix, got_of_point):
    from tests.helpers import check_matrix_summary
    n = fix["n"]
    for k, pt in enumerate(fix["points"]):
        got = got_of_point(k)
        for key in ("rhs", "adj", "quad"):
            want = np.array(pt[key], float)
            scale = float(np.max(np.abs(want))) if want.size else 0.0
            np.testing.assert_allclose(np.asarray(got[key]).ravel(), want, rtol=1e-13, atol=64 * 2.3e-16 * scale, err_msg=key)
        # (oracle.eval / eval_callbacks hand the matrices back as M[i, j], row = output, like the fixture)
        check_matrix_summary(np.asarray(got["jac"]).reshape(n, n), pt["jac"])
        check_matrix_summary(np.asarray(got["adjjac"]).reshape(n, n), pt["adjjac"])
        assert np.asarray(got["codes"]).tolist() == pt["codes"]


@pytest.mark.parametrize("name", PINNED)
def test_sweep_callbacks_match_the_reference_pipeline(name, golden_dir):
    """CPU: the generated C of five sweep shapes (host build = the text the kernels include) vs values the
    REFERENCE's symbolic pipeline produced for the same model -- the sweep's callbacks are no longer code generator
    against itself (missing #7)."""
    fix, prob, t, y, lam, ps, pr = _pinned_points(name, golden_dir)
    orc = make_oracle(name)
    _check_against_reference(fix, lambda k: orc.eval(t[k], y[k], lam[k], ps[k], pr[k]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", PINNED)
def test_device_sweep_callbacks_match_the_reference_pipeline(name, golden_dir):
    """... and the DEVICE functions of the same shapes (every kernel family's callback staging), equal to the host
    build bit for bit."""
    from sunode_amd.solver import Solver
    fix, prob, t, y, lam, ps, pr = _pinned_points(name, golden_dir)
    eng = Solver(prob)._engine()
    # (the engine-level call takes the remainder vector the generated source reads: hoisted / packed copies appended)
    got = eng.eval_callbacks(t, y, lam, ps, np.array([prob.extend_remainder(row) for row in pr]))
    _check_against_reference(fix, lambda k: {key: got[key][k] for key in ("rhs", "jac", "adj", "quad", "adjjac", "codes")})
    orc = make_oracle(name)
    for k in range(len(t)):
        host = orc.eval(t[k], y[k], lam[k], ps[k], pr[k])
        for key in ("rhs", "jac", "adj", "quad", "adjjac"):
            np.testing.assert_array_equal(np.asarray(got[key][k]).ravel(), np.asarray(host[key]).ravel())


@pytest.mark.parametrize("name", ["lv12", "rn12_4"])
def test_oracle_gradients_of_sweep_shapes_match_truth(name, golden_dir):
    """DOP853 truth (tools/make_golden_truth.py --sweep) for two sweep shapes: the oracle's states / gradients at the
    SURVEY 8(c) bars -- the sweep's gradients were only asserted finite and non-zero."""
    d = np.load(os.path.join(golden_dir, "truth_sweep_%s.npz" % name))
    orc = make_oracle(name)
    cfg = orc.config(rtol=1e-8, atol=1e-8, rtolB=1e-8, atolB=1e-8, rtolQB=1e-8, atolQB=1e-8)
    tv = d["tvals"]
    y, st, _ = orc.solve_forward(cfg, d["y0"], d["ps"], d["pr"], float(d["t0"]), tv, nthreads=4)
    g, lam, stb, _ = orc.solve_backward(cfg, tv[-1], float(d["t0"]), tv, d["grads"], nthreads=4)
    assert (st == 0).all() and (stb == 0).all()
    _truth_bars(y, g, lam, d)


def _truth_bars(y, g, lam, d):
    k = len(d["y_out"])
    assert np.max(np.abs(y[:k] - d["y_out"]) / np.abs(d["y_out"]).max(axis=(0, 1))) < 1e-5
    assert np.max(np.abs(g[:k] - d["grad_params"]) / np.abs(d["grad_params"]).max(axis=1, keepdims=True)) < 4e-6
    assert np.max(np.abs(-lam[:k] - d["grad_y0"]) / np.abs(d["grad_y0"]).max(axis=1, keepdims=True)) < 4e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name,B", ADJOINT_CASES, ids=[c[0] for c in ADJOINT_CASES])
def test_default_mapping_adjoint_equals_oracle(name, B, golden_dir):
    from sunode_amd.solver import AdjointSolver
    assert not os.environ.get("SA_FORCE_GROUP")
    prob = make_problem(name)
    d = batch_of(name, B)
    tol = dict(abstol=d["atol"], reltol=d["rtol"], backward_abstol=d["atol"], backward_reltol=d["rtol"],
               quad_abstol=d["atol"], quad_reltol=d["rtol"])
    # batch_mapping="fixed": the family kernel_variant selects for (n, p) is what this sweep is about (the small-batch
    # switch of 4- / 5-state models to lane groups has its own test below)
    sol = AdjointSolver(prob, batch_mapping="fixed", **tol)
    tv = d["tvals"]
    y, st, stats = sol.solve_forward_batch(d["t0"], tv, d["y0"], d["ps"], d["pr"])
    g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], d["t0"], tv, d["grads"])
    orc = make_oracle(name)
    cfg = orc.config(rtol=d["rtol"], atol=d["atol"], rtolB=d["rtol"], atolB=d["atol"], rtolQB=d["rtol"],
                     atolQB=d["atol"])
    yo, so, sto = orc.solve_forward(cfg, d["y0"], d["ps"], d["pr"], d["t0"], tv, nthreads=8)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], d["t0"], tv, d["grads"], nthreads=8)
    assert (so == 0).all() and (sbo == 0).all()
    np.testing.assert_array_equal(st, so)
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(stb, sbo)
    np.testing.assert_array_equal(statsb[:, CMP_B], stbo[:, CMP_B])
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)
    assert np.isfinite(g).all() and np.abs(g).max() > 0
    truth = os.path.join(golden_dir, "truth_sweep_%s.npz" % name)
    if os.path.exists(truth):           # lv12, rn12_4: the first draws of the batch against DOP853 truth
        _truth_bars(y, g, lam, np.load(truth))


@pytest.mark.gpu
def test_small_batches_of_a_five_state_model_run_in_lane_groups():
    """``batch_mapping="auto"`` (the default): a model the engine maps to one lane per instance with n >= 4 states runs
    with 16 / 8 / 4 lanes per instance while a handle's batch is <= 4 096 / 8 192 / 16 384 (profiles/
    r06_mapping_by_batch.txt, r06_lanes_small_models.txt: +40 ... 75 % for n = 5, p = 8), in the one-lane kernel above -- one solver object, both code objects, every result equal to the oracle's
    and to the fixed mapping's."""
    from sunode_amd import _native
    from sunode_amd.solver import AdjointSolver
    assert not os.environ.get("SA_FORCE_GROUP")
    name = "rn5_8"
    prob = make_problem(name)
    assert _native.small_batch_group(prob.native_source(), batch=12000) == "wave4"
    assert _native.small_batch_group(make_problem("lv12").native_source(), batch=20000) is None
    d = batch_of(name, 20000)
    tol = dict(abstol=d["atol"], reltol=d["rtol"], backward_abstol=d["atol"], backward_reltol=d["rtol"],
               quad_abstol=d["atol"], quad_reltol=d["rtol"])
    auto, fixed = AdjointSolver(prob, **tol), AdjointSolver(prob, batch_mapping="fixed", **tol)
    tv = d["tvals"]
    orc = make_oracle(name)
    cfg = orc.config(rtol=d["rtol"], atol=d["atol"], rtolB=d["rtol"], atolB=d["atol"], rtolQB=d["rtol"], atolQB=d["atol"])
    for B, family in ((70, ("bdf_wave.hip", 16)), (20000, ("bdf_kernels.hip", 1)), (6000, ("bdf_wave.hip", 8)),
                      (12000, ("bdf_wave.hip", 4)), (300, ("bdf_wave.hip", 16))):
        res = {}
        for tag, sol in (("auto", auto), ("fixed", fixed)):
            y, st, stats = sol.solve_forward_batch(d["t0"], tv, d["y0"][:B], d["ps"][:B], d["pr"])
            assert sol._engine().variant == (family if tag == "auto" else ("bdf_kernels.hip", 1))
            g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], d["t0"], tv, d["grads"][:B])
            res[tag] = (y, st, stats[:, CMP], g, lam, stb, statsb[:, CMP_B])
        for a, b in zip(res["auto"], res["fixed"]):
            np.testing.assert_array_equal(a, b)
        k = min(B, 512)
        yo, so, sto = orc.solve_forward(cfg, d["y0"][:k], d["ps"][:k], d["pr"], d["t0"], tv, nthreads=8)
        go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], d["t0"], tv, d["grads"][:k], nthreads=8)
        np.testing.assert_array_equal(res["auto"][0][:k], yo)
        np.testing.assert_array_equal(res["auto"][3][:k], go)
        np.testing.assert_array_equal(res["auto"][4][:k], lo)
        np.testing.assert_array_equal(res["auto"][6][:k], stbo[:, CMP_B])


@pytest.mark.gpu
def test_small_batches_of_a_lane_group_model_get_more_lanes_per_instance():
    """SEIR (16 states, 4 lanes per instance at BASELINE's batch): 16 lanes per instance up to 4 096 instances on a
    handle, 8 up to 8 192 (profiles/r06_lanes_by_batch.txt: +25 ... 37 %), the 4-lane groups above -- chosen per call
    by ``AdjointSolver``, all equal to the oracle."""
    from sunode_amd.solver import AdjointSolver
    from tools.problems import seir_batch
    assert not os.environ.get("SA_FORCE_GROUP")
    prob = make_problem("seir")
    d = seir_batch(9000)
    tv = d["tvals"][::5]
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(16)[None, :])
    tol = dict(abstol=1e-8, reltol=1e-8, backward_abstol=1e-8, backward_reltol=1e-8, quad_abstol=1e-8, quad_reltol=1e-8)
    sol = AdjointSolver(prob, **tol)
    orc = make_oracle("seir")
    cfg = orc.config(rtol=1e-8, atol=1e-8, rtolB=1e-8, atolB=1e-8, rtolQB=1e-8, atolQB=1e-8)
    k = 200
    yo, so, sto = orc.solve_forward(cfg, d["y0"][:k], d["ps"][:k], d["pr"], 0.0, tv, nthreads=8)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=8)
    for B, lanes in ((70, 16), (5000, 8), (9000, 4), (k, 16)):
        y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"][:B], d["ps"][:B], d["pr"])
        assert sol._engine().variant == ("bdf_wave.hip", lanes)
        g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
        m = min(B, k)
        assert (st == 0).all() and (stb == 0).all()
        np.testing.assert_array_equal(stats[:m, CMP], sto[:m, CMP])
        np.testing.assert_array_equal(y[:m], yo[:m])
        np.testing.assert_array_equal(statsb[:m, CMP_B], stbo[:m, CMP_B])
        np.testing.assert_array_equal(g[:m], go[:m])
        np.testing.assert_array_equal(lam[:m], lo[:m])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["simultaneous", "staggered"])
@pytest.mark.parametrize("name,B", SENS_CASES, ids=[c[0] for c in SENS_CASES])
def test_default_mapping_sensitivities_equal_oracle(name, B, mode):
    from sunode_amd.solver import Solver
    assert not os.environ.get("SA_FORCE_GROUP")
    prob = make_problem(name)
    d = batch_of(name, B)
    sol = Solver(prob, abstol=d["atol"], reltol=d["rtol"], sens_mode=mode)
    n, p = prob.n_states, prob.n_params
    sens0 = np.zeros((p, n))
    y, s, st, stats = sol.solve_sens_batch(d["t0"], d["tvals"], d["y0"], d["ps"], d["pr"], sens0)
    orc = make_oracle(name)
    cfg = orc.config(rtol=d["rtol"], atol=d["atol"])
    yo, so_, sto_, statso = orc.solve_sens(cfg, d["y0"], d["ps"], d["pr"], sens0, d["t0"], d["tvals"], mode=mode,
                                           nthreads=8)
    assert (sto_ == 0).all()
    np.testing.assert_array_equal(st, sto_)
    np.testing.assert_array_equal(stats[:, CMP[:8]], statso[:, CMP[:8]])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(s, so_)
    assert np.abs(s).max() > 0
