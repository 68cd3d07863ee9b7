"""The differential guard (include/sunode_amd.h: sa_solver_attach_guard; sunode_amd/_native.py NativeSolver).

Every user model is a new code object (the reference accepts any sympy system, /root/reference/sunode/symode/
problem.py:25-33) compiled by a back end that round 4 caught miscompiling this source family once
(profiles/r04_sens_anomaly.txt), and a GPU box has no CPU oracle.  The product therefore runs the first instances of
the first batch through the default AND the conservative build and compares them bit for bit on the device.  Here:
the bookkeeping (CPU), a clean model passing, both mismatch paths with builds that differ on purpose, and the
known-bad compiler combination of round 4 being caught and repaired.  (The rest of the suite runs with SA_GUARD=0 --
tests/conftest.py -- because it compares everything with the oracle anyway and would compile every test model twice.)
"""
import json
import os
import warnings

import numpy as np
import pytest

from tests.helpers import make_oracle, make_problem
from tools.problems import lv_batch, seir_batch


def test_conservative_build_is_a_separate_code_object(monkeypatch):
    from sunode_amd import _native
    src = make_problem("lv").native_source()
    fast, safe = _native.code_object_path(src), _native.code_object_path(src, safe=True)
    assert fast != safe
    assert _native.code_object_path(src, compact=True, safe=True) not in (fast, safe)
    monkeypatch.setenv("SA_VGPR_LIVERANGE_OPT", "0")          # everything conservative: one build, no guard
    assert _native.code_object_path(src) == _native.code_object_path(src, safe=True)
    assert not _native.guard_enabled()
    monkeypatch.delenv("SA_VGPR_LIVERANGE_OPT")
    monkeypatch.setenv("SA_GUARD", "1")
    assert _native.guard_enabled()
    monkeypatch.setenv("SA_GUARD", "0")
    assert not _native.guard_enabled()


def test_verdict_file_round_trip(tmp_path, monkeypatch):
    from sunode_amd import _native
    fast, safe = str(tmp_path / "sa_aaaa.hsaco"), str(tmp_path / "sa_bbbb.hsaco")
    assert _native.read_guard_verdict(fast, safe) == {}
    kinds = {"adjoint": {"verdict": "identical", "n_sample": 64}}
    _native.write_guard_verdict(fast, safe, kinds, "")
    assert _native.read_guard_verdict(fast, safe) == kinds
    assert _native.read_guard_verdict(fast, str(tmp_path / "sa_cccc.hsaco")) == {}      # another partner build
    doc = json.load(open(_native.guard_verdict_path(fast)))
    doc["toolchain"] = "other"
    json.dump(doc, open(_native.guard_verdict_path(fast), "w"))
    assert _native.read_guard_verdict(fast, safe) == {}                                  # another toolchain


def _lv(B):
    prob = make_problem("lv")
    d = lv_batch(B)
    return prob, d, d["params"][:, prob.params_subset.subset_index], d["params"][:, prob.params_subset.remainder_index]


TOL = dict(abstol=1e-8, reltol=1e-8, backward_abstol=1e-8, backward_reltol=1e-8, quad_abstol=1e-8, quad_reltol=1e-8)


def _forget_verdict(src, **kw):
    from sunode_amd import _native
    path = _native.guard_verdict_path(_native.code_object_path(src, **kw))
    if os.path.exists(path):
        os.remove(path)
    return path


@pytest.mark.gpu
def test_guard_verifies_a_clean_model_and_keeps_the_verdict(monkeypatch):
    """LV through the product API with the guard on: the first forward + backward pair checks 64 instances on the
    device, the results are the oracle's, the verdict lands next to the code object and the next solver object
    starts with nothing pending."""
    from sunode_amd.solver import AdjointSolver, Solver
    monkeypatch.setenv("SA_GUARD", "1")
    prob, d, ps, pr = _lv(300)
    verdict = _forget_verdict(prob.native_source())
    sol = AdjointSolver(prob, **TOL)
    eng = sol._engine()
    assert eng.guard_report["enabled"] and eng.guard_state()["pending"] == ["adjoint"]     # the kind this handle runs
    tv = d["tvals"]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
        assert "adjoint" in eng.guard_state()["pending"]            # the backward pass finishes the check
        g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, np.ones((len(tv), 2)))
    state = eng.guard_state()
    assert state["verified"] == ["adjoint"] and not state["differs"] and not state["using_conservative"]
    assert state["n_sample"]["adjoint"] == 64
    orc = make_oracle("lv")
    cfg = orc.config(rtol=1e-8, atol=1e-8, rtolB=1e-8, atolB=1e-8, rtolQB=1e-8, atolQB=1e-8)
    yo, so, sto = orc.solve_forward(cfg, d["y0"], ps, pr, 0.0, tv, nthreads=8)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], 0.0, tv, np.ones((len(tv), 2)), nthreads=8)
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)
    doc = json.load(open(verdict))
    assert doc["kinds"]["adjoint"] == {"verdict": "identical", "n_sample": 64}
    sol2 = AdjointSolver(prob, **TOL)
    assert "adjoint" not in sol2._engine().guard_state()["pending"]
    # the plain forward solve of the same code object is its own kind; a batch of 5 is checked, and checked again
    plain = Solver(prob, abstol=1e-8, reltol=1e-8)
    e3 = plain._engine()
    for _ in range(3):
        assert "plain" in e3.guard_state()["pending"]
        plain.solve_batch(0.0, tv, d["y0"][:5], ps[:5], pr[:5])
    assert e3.guard_state()["verified"] == ["plain"] and e3.guard_state()["n_sample"]["plain"] == 5
    assert not e3.guard_state()["pending"]
    # a verdict from batches of five stays with the process: nothing about "plain" on disk
    assert "plain" not in json.load(open(verdict))["kinds"]


@pytest.mark.gpu
def test_guard_finds_no_difference_on_any_shape_of_the_sweep(monkeypatch):
    """Default build == conservative build, bit for bit on the device, for every shape of tests/test_shape_sweep.py
    (every kernel family, both sides of every mapping boundary): the guard verifies all 29 models -- no false alarm,
    and no default mapping depends on the pass round 4 caught."""
    from sunode_amd.solver import AdjointSolver
    from tools.sweep_cases import ADJOINT_CASES, batch_of
    monkeypatch.setenv("SA_GUARD", "1")
    seen = {}
    with warnings.catch_warnings():
        warnings.simplefilter("error")                  # a RuntimeWarning of the guard fails the test
        for name, B in ADJOINT_CASES:
            B = min(B, 80)              # (the sweep's own test runs the large batches; here: default vs conservative build)
            prob = make_problem(name)
            d = batch_of(name, B)
            src = prob.native_source()
            from sunode_amd import _native
            _forget_verdict(src, compact=_native.default_compact_trajectory(src))
            tol = dict(abstol=d["atol"], reltol=d["rtol"], backward_abstol=d["atol"], backward_reltol=d["rtol"],
                       quad_abstol=d["atol"], quad_reltol=d["rtol"])
            sol = AdjointSolver(prob, batch_mapping="fixed", **tol)     # (the family of the shape, as in the sweep)
            tv = d["tvals"]
            for _ in range(3 if B < 16 else 1):         # (a check with fewer than 16 instances is repeated twice)
                sol.solve_forward_batch(d["t0"], tv, d["y0"], d["ps"], d["pr"])
                sol.solve_backward_batch(tv[-1], d["t0"], tv, d["grads"])
            st = sol._engine().guard_state()
            seen[name] = (st["verified"], st["differs"], st["n_sample"]["adjoint"])
            del sol
    assert all(v == (["adjoint"], [], min(B, 64)) for (name, B), v in zip(ADJOINT_CASES, seen.values())), seen
    assert len(seen) == len(ADJOINT_CASES)


@pytest.mark.gpu
def test_guard_under_several_handles_of_one_solver(monkeypatch):
    """AdjointSolver(devices=[0, 0, 0]): every handle carries its own guard and reports from its own host thread; the
    three checks agree, the verdict file stays valid JSON, the results are the one-handle results."""
    from sunode_amd.solver import AdjointSolver
    monkeypatch.setenv("SA_GUARD", "1")
    prob, d, ps, pr = _lv(600)
    verdict = _forget_verdict(prob.native_source())
    tv = d["tvals"]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        sol = AdjointSolver(prob, devices=[0, 0, 0], interleaved=True, arena_gib=6, **TOL)
        y, st, _ = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
        g, lam, stb, _ = sol.solve_backward_batch(tv[-1], 0.0, tv, np.ones((len(tv), 2)))
    for eng in sol._engines():
        state = eng.guard_state()
        assert state["verified"] == ["adjoint"] and not state["differs"] and state["n_sample"]["adjoint"] == 64
    assert json.load(open(verdict))["kinds"]["adjoint"]["verdict"] == "identical"
    monkeypatch.setenv("SA_GUARD", "0")
    one = AdjointSolver(prob, **TOL)
    y1, _, _ = one.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
    g1, lam1, _, _ = one.solve_backward_batch(tv[-1], 0.0, tv, np.ones((len(tv), 2)))
    np.testing.assert_array_equal(y, y1)
    np.testing.assert_array_equal(g, g1)
    np.testing.assert_array_equal(lam, lam1)


def _attach_other_build(monkeypatch, prob, define):
    """A NativeSolver on the default LV build whose guard partner is a build that differs ON PURPOSE."""
    from sunode_amd import _native
    src = prob.native_source()
    monkeypatch.setenv("SA_KERNEL_DEFINES", define)
    other = _native.build_code_object(src)
    monkeypatch.delenv("SA_KERNEL_DEFINES")
    eng = _native.NativeSolver(src, n_states=prob.n_states, guard=False, rtol=1e-8, atol=1e-8, rtolB=1e-8, atolB=1e-8,
                               rtolQB=1e-8, atolQB=1e-8)
    eng._check(eng.L.sa_solver_attach_guard(eng._h, other.encode(), 0, 0))
    return eng, other


def _run_pair(eng, d, ps, pr, B):
    from sunode_amd import _native
    tv = np.ascontiguousarray(d["tvals"])
    y = np.zeros((B, len(tv), 2)); st = np.zeros(B, np.int32); stats = np.zeros((B, 16), np.int64)
    eng.solve(_native.SA_MEM_HOST, B, np.ascontiguousarray(d["y0"]), np.ascontiguousarray(ps),
              np.ascontiguousarray(pr), pr.shape[1], 0.0, tv, len(tv), y, st, stats, adjoint=True)
    g = np.zeros((B, 2)); lam = np.zeros((B, 2)); stb = np.zeros(B, np.int32); statsb = np.zeros((B, 16), np.int64)
    grads = np.ones((len(tv), 2))
    mid = eng.guard_state()
    eng.solve_backward(_native.SA_MEM_HOST, B, np.ascontiguousarray(ps), np.ascontiguousarray(pr), pr.shape[1],
                       tv[-1], 0.0, tv, len(tv), grads, 0, g, lam, stb, statsb)
    return y, g, lam, mid


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["FORWARD", "BACKWARD"])
def test_guard_switches_to_the_other_build_on_any_difference(where, monkeypatch):
    """The C library's two mismatch paths, driven with a partner build that adds 1.0 to one output on purpose:
    a difference in the forward pass switches and launches the batch again with the partner build before the call
    returns; a difference that only shows in the backward pass switches there and repeats the batch's forward pass
    with the partner build before integrating backward.  Afterwards the handle behaves exactly like a handle
    created on the partner build."""
    from sunode_amd import _native
    monkeypatch.setenv("SA_GUARD", "0")
    prob, d, ps, pr = _lv(200)
    eng, other = _attach_other_build(monkeypatch, prob, "-DSA_TEST_PERTURB_" + where)
    y, g, lam, mid = _run_pair(eng, d, ps, pr, 200)
    st = eng.guard_state()
    assert st["differs"] == ["adjoint"] and st["using_conservative"] and not st["pending"]
    assert ("forward pass" if where == "FORWARD" else "backward pass") in st["detail"]
    assert (mid["differs"] == ["adjoint"]) == (where == "FORWARD")
    # reference: the unperturbed engine
    ref = _native.NativeSolver(prob.native_source(), n_states=2, guard=False, rtol=1e-8, atol=1e-8, rtolB=1e-8,
                               atolB=1e-8, rtolQB=1e-8, atolQB=1e-8)
    y0_, g0_, lam0_, _ = _run_pair(ref, d, ps, pr, 200)
    if where == "FORWARD":
        np.testing.assert_array_equal(y[:, -1, 0], y0_[:, -1, 0] + 1.0)      # launched AFTER the switch
        np.testing.assert_array_equal(lam, lam0_)
    else:
        np.testing.assert_array_equal(y, y0_)                                # the forward kernels are the same code
        np.testing.assert_array_equal(lam[:, 0], lam0_[:, 0] + 1.0)          # every instance, not just the sample
    np.testing.assert_array_equal(g, g0_)
    # later batches stay on the partner build, without further checks
    y2, g2, lam2, _ = _run_pair(eng, d, ps, pr, 70)
    np.testing.assert_array_equal(lam2[:, 0] if where == "BACKWARD" else y2[:, -1, 0],
                                  (lam0_[:70, 0] if where == "BACKWARD" else y0_[:70, -1, 0]) + 1.0)


@pytest.mark.gpu
def test_guard_catches_the_round4_miscompile(monkeypatch):
    """profiles/r04_sens_anomaly.txt: SEIR's 4-lane forward-sensitivity kernel with the coefficient vectors parked in
    LDS (-DSA_SENS_CTL_PARK) and -disable-machine-licm is miscompiled by SIOptimizeVGPRLiveRange (half of the step
    counters wrong).  With that combination forced as the DEFAULT build, the product notices on its first batch
    (RuntimeWarning), switches to the conservative build and returns the oracle's results."""
    from sunode_amd import _native
    from sunode_amd.solver import Solver
    monkeypatch.setenv("SA_GUARD", "1")
    monkeypatch.setenv("SA_KERNEL_DEFINES", "-DSA_SENS_CTL_PARK -DSA_SENS_UNROLL")     # (+ the round-4/5 form of the parameter loop:
    # with the loop kept a loop -- round 6 -- this combination no longer trips the pass)
    monkeypatch.setenv("SA_CLANG_FLAGS", "-mllvm -disable-machine-licm")
    prob = make_problem("seir")
    assert _native.kernel_variant(prob.native_source(), sens=True) == ("bdf_wave.hip", 4)
    _forget_verdict(prob.native_source(), sens=True)
    B = 21
    d = seir_batch(B)
    tv = d["tvals"][::5]
    sens0 = np.zeros((prob.n_params, prob.n_states))
    sens0[0, 3] = 0.5
    sol = Solver(prob, abstol=1e-8, reltol=1e-8, sens_mode="simultaneous")
    with pytest.warns(RuntimeWarning, match="differential guard"):
        y, S, status, stats = sol.solve_sens_batch(0.0, tv, d["y0"], d["ps"], d["pr"], sens0)
    state = sol._engine().guard_state()
    assert state["differs"] == ["sens"] and state["using_conservative"]
    orc = make_oracle("seir")
    yo, So, so, sto = orc.solve_sens(orc.config(rtol=1e-8, atol=1e-8), d["y0"], d["ps"], d["pr"], sens0, 0.0, tv,
                                     mode="simultaneous", nthreads=8)
    cmp = [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13]
    np.testing.assert_array_equal(stats[:, cmp], sto[:, cmp])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(S, So)
    # the next solver object of this process (same builds) starts on the conservative build right away
    with pytest.warns(RuntimeWarning, match="different results in an earlier run"):
        sol2 = Solver(prob, abstol=1e-8, reltol=1e-8, sens_mode="simultaneous")
        eng2 = sol2._engine()
    assert eng2.guard_report["using_conservative"] and not eng2.guard_state()["pending"]
    y2, S2, _, _ = sol2.solve_sens_batch(0.0, tv, d["y0"], d["ps"], d["pr"], sens0)
    np.testing.assert_array_equal(S2, So)


@pytest.mark.gpu
def test_guard_sample_is_chosen_from_the_batch_statistics(monkeypatch):
    """A partner build that differs ONLY for instances with the batch's largest number of error-test failures, none of
    which is among the first 64 draws: the round-5 guard (first 64 instances) could not see it; the sample chosen
    from the batch's own counters does, and the batch the caller receives is the partner build's."""
    from sunode_amd import _native
    monkeypatch.setenv("SA_GUARD", "0")
    B = 3000
    prob, d, ps, pr = _lv(B)
    ref = _native.NativeSolver(prob.native_source(), n_states=2, guard=False, rtol=1e-8, atol=1e-8, rtolB=1e-8,
                               atolB=1e-8, rtolQB=1e-8, atolQB=1e-8)
    tv = np.ascontiguousarray(d["tvals"])
    y0_ = np.zeros((B, len(tv), 2)); st0 = np.zeros(B, np.int32); stats0 = np.zeros((B, 16), np.int64)
    ref.solve(_native.SA_MEM_HOST, B, np.ascontiguousarray(d["y0"]), np.ascontiguousarray(ps), np.ascontiguousarray(pr),
              pr.shape[1], 0.0, tv, len(tv), y0_, st0, stats0, adjoint=True)
    netf = stats0[:, 6]
    T = int(netf.max())
    rare = np.flatnonzero(netf >= T)
    assert T >= 2 and 0 < len(rare) <= 8 and rare.min() >= 64, (T, rare)       # rare, and outside the old prefix sample
    eng, other = _attach_other_build(monkeypatch, prob, "-DSA_TEST_PERTURB_NETF=%d" % T)
    y = np.zeros((B, len(tv), 2)); st = np.zeros(B, np.int32); stats = np.zeros((B, 16), np.int64)
    eng.solve(_native.SA_MEM_HOST, B, np.ascontiguousarray(d["y0"]), np.ascontiguousarray(ps), np.ascontiguousarray(pr),
              pr.shape[1], 0.0, tv, len(tv), y, st, stats, adjoint=True)
    state = eng.guard_state()
    assert state["differs"] == ["adjoint"] and state["using_conservative"]
    inst = int(state["detail"].split("of instance ")[1].split()[0])
    assert inst in rare                                    # the message names the batch index, not the sample row
    want = y0_.copy()
    want[rare, -1, 0] += 1.0
    np.testing.assert_array_equal(y, want)                 # the batch was launched again with the partner build
    # a prefix-only batch (B <= sample size) is used in place: same verdict machinery, nothing to gather
    eng2, _ = _attach_other_build(monkeypatch, prob, "-DSA_TEST_PERTURB_NETF=%d" % T)
    k = 40
    y2 = np.zeros((k, len(tv), 2)); st2 = np.zeros(k, np.int32); stats2 = np.zeros((k, 16), np.int64)
    eng2.solve(_native.SA_MEM_HOST, k, np.ascontiguousarray(d["y0"][:k]), np.ascontiguousarray(ps[:k]),
               np.ascontiguousarray(pr[:k]), pr.shape[1], 0.0, tv, len(tv), y2, st2, stats2, adjoint=True)
    assert not eng2.guard_state()["differs"]
    np.testing.assert_array_equal(y2, y0_[:k])


@pytest.mark.gpu
def test_guard_rechecks_a_later_batch_with_a_status_the_sample_never_had(monkeypatch):
    """Verified on a clean first batch; a later host-memory batch contains a FAILING instance (a status code the
    verified sample never showed): the kind is checked once more on that batch (the failing instance is in the
    sample), stays verified, and the window closes."""
    from sunode_amd.solver import AdjointSolver
    monkeypatch.setenv("SA_GUARD", "1")
    prob, d, ps, pr = _lv(400)
    _forget_verdict(prob.native_source())
    tv = d["tvals"]
    sol = AdjointSolver(prob, mxsteps=40, **TOL)
    eng = sol._engine()
    g1 = np.ones((len(tv), 2))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
        sol.solve_backward_batch(tv[-1], 0.0, tv, g1)
        s1 = eng.guard_state()
        assert s1["verified"] == ["adjoint"] and s1["recheck_open"] and not s1["pending"]
        checks_before = eng.guard_report["kinds"]["adjoint"]["n_sample"]
        y0 = d["y0"].copy()
        y0[333] = [np.nan, 1.0]                           # this instance fails at once (non-finite rhs)
        y, st, _ = sol.solve_forward_batch(0.0, tv, y0, ps, pr)
        assert st[333] != 0 and (np.delete(st, 333) == 0).all()
        s2 = eng.guard_state()
        assert s2["pending"] == ["adjoint"] and not s2["recheck_open"]         # re-opened, the backward pass finishes it
        sol.solve_backward_batch(tv[-1], 0.0, tv, g1)
    s3 = eng.guard_state()
    assert s3["verified"] == ["adjoint"] and not s3["pending"] and not s3["recheck_open"] and not s3["differs"]
    assert checks_before == 64 and eng._guard_open is False


@pytest.mark.gpu
def test_forward_only_use_of_the_adjoint_solver_stops_paying_for_the_guard(monkeypatch):
    """solve_forward_batch without a backward call cannot finish the adjoint check: after two such calls the shadows
    are dropped and later forward calls only enqueue; the check resumes with the first forward call after a backward
    call."""
    from sunode_amd.solver import AdjointSolver
    monkeypatch.setenv("SA_GUARD", "1")
    prob, d, ps, pr = _lv(200)
    _forget_verdict(prob.native_source())
    tv = d["tvals"]
    sol = AdjointSolver(prob, **TOL)
    eng = sol._engine()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for _ in range(5):
            y, st, _ = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
            assert (st == 0).all() and eng.guard_state()["pending"] == ["adjoint"]
        sol.solve_backward_batch(tv[-1], 0.0, tv, np.ones((len(tv), 2)))        # no forward check in flight: nothing booked
        assert eng.guard_state()["pending"] == ["adjoint"]
        sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
        sol.solve_backward_batch(tv[-1], 0.0, tv, np.ones((len(tv), 2)))
    assert eng.guard_state()["verified"] == ["adjoint"] and eng.guard_state()["n_sample"]["adjoint"] == 64


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["lv_fullsize", "tiled_arena", "two_handles_failing_shard", "constraints", "hermite"])
def test_product_configuration_guard_on(case, monkeypatch):
    """The configuration users run (guard ON, everything else default) through the paths the rest of the suite covers
    with SA_GUARD=0: BASELINE config 2 at full size, tiled re-integration, two handles with a failing instance in one
    shard, constraints, Hermite interpolation -- results equal the guard-off run bit for bit, the guard verifies, no
    warning."""
    from sunode_amd.solver import AdjointSolver
    from tools.problems import robertson_batch
    kw, B, name = dict(TOL), 2000, "lv"
    if case == "lv_fullsize":
        B = 65536
    elif case == "tiled_arena":
        name, B = "robertson", 4096
        kw = dict(abstol=1e-10, reltol=1e-8, backward_abstol=1e-10, backward_reltol=1e-8, quad_abstol=1e-10,
                  quad_reltol=1e-8, arena_gib=0.05)
    elif case == "two_handles_failing_shard":
        kw.update(devices=[0, 0], arena_gib=4)
    elif case == "constraints":
        kw.update(constraints=np.array([1.0, 1.0]))
    elif case == "hermite":
        kw.update(interpolation="hermite")
    prob = make_problem(name)
    if name == "lv":
        d = lv_batch(B)
        y0, ps, pr = d["y0"].copy(), d["params"][:, :2], d["params"][:, 2:]
    else:
        d = robertson_batch(B)
        y0, ps, pr = d["y0"], d["params"], np.zeros(0)
    if case == "two_handles_failing_shard":
        y0[B - 7] = [np.nan, 1.0]
    tv = d["tvals"]
    n = prob.n_states
    grads = 1.0 + 0.5 * np.cos(1.7 * np.arange(len(tv))[:, None] + 0.9 * np.arange(n)[None, :])
    results = {}
    for guard in ("0", "1"):
        monkeypatch.setenv("SA_GUARD", guard)
        if guard == "1":
            from sunode_amd import _native
            src = prob.native_source()
            _forget_verdict(src, constraints=case == "constraints", hermite=case == "hermite",
                            compact=_native.default_compact_trajectory(src, case == "hermite"))
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            sol = AdjointSolver(prob, **kw)
            y, st, stats = sol.solve_forward_batch(0.0, tv, y0, ps, pr)
            g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
        if guard == "1":
            for eng in sol._engines():
                state = eng.guard_state()
                assert state["verified"] == ["adjoint"] and not state["differs"] and state["n_sample"]["adjoint"] == 64
            if case == "tiled_arena":
                assert sol._engine().arena_info()[2]          # the batch was re-integrated tile by tile
        results[guard] = (y, st, stats[:, :9], g, lam, stb, statsb[:, :13])
        for eng in sol._engines():
            eng.close()
    for a, b in zip(results["0"], results["1"]):
        np.testing.assert_array_equal(a, b)
    if case == "two_handles_failing_shard":
        assert results["1"][1][B - 7] != 0 and (np.delete(results["1"][1], B - 7) == 0).all()
