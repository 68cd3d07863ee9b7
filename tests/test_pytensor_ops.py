"""The pytensor Ops: numeric halves (``perform``) against the solvers they wrap, and graph construction
(``solve_ivp``) + ``grad`` wiring executed end to end against the CPU oracle's gradients.

pytensor is not installed in this image, so a stub package (tests/stubs/pytensor: a tiny lazy graph with the
tensor vocabulary the wrapper module uses) stands in for it; with the real pytensor installed the ``perform``
tests run against it unchanged (the graph tests use the stub's ``evaluate`` and skip otherwise)."""
import importlib
import os
import sys

import numpy as np
import pytest

from tests.helpers import make_problem

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ops():
    try:
        import pytensor  # noqa: F401
        stub = None
    except ImportError:
        stub = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")
        sys.path.insert(0, stub)
    sys.modules.pop("sunode_amd.wrappers.as_pytensor", None)
    mod = importlib.import_module("sunode_amd.wrappers.as_pytensor")
    yield mod
    if stub:
        sys.path.remove(stub)
        for name in [n for n in sys.modules if n == "pytensor" or n.startswith("pytensor.")]:
            del sys.modules[name]
        sys.modules.pop("sunode_amd.wrappers.as_pytensor", None)


def _run(op, inputs, n_out):
    outputs = [[None] for _ in range(n_out)]
    op.perform(None, inputs, outputs)
    return [o[0] for o in outputs]


def test_adjoint_ops_perform(ops):
    from sunode_amd.solver import AdjointSolver
    prob = make_problem("lv")
    tol = 1e-9
    solver = AdjointSolver(prob, abstol=tol, reltol=tol, backward_abstol=tol, backward_reltol=tol,
                           quad_abstol=tol, quad_reltol=tol)
    y0 = np.array([1.0, 0.1]); params = np.array([0.1, 0.2]); fixed = np.array([0.3, 0.4])
    tv = np.linspace(0, 10, 21)
    grads = np.cos(np.arange(42.0)).reshape(21, 2)
    y, = _run(ops.SolveODEAdjoint(solver), [y0, params, fixed, np.array(0.0), tv], 1)
    lam, grad = _run(ops.SolveODEAdjointBackward(solver), [y0, params, fixed, grads, np.array(0.0), tv], 2)
    yd, _, _ = solver.solve_forward_batch(0.0, tv, y0[None], params[None], fixed)
    gd, ld, _, _ = solver.solve_backward_batch(tv[-1], 0.0, tv, grads)
    np.testing.assert_array_equal(y, yd[0])
    np.testing.assert_array_equal(grad, gd[0])
    np.testing.assert_array_equal(lam, ld[0])
    # batched pair: leading axis = parameter draw
    Y0 = np.stack([y0, y0 * 1.1]); P = np.stack([params, params * 0.9])
    yb, = _run(ops.SolveODEAdjointBatch(solver), [Y0, P, fixed, np.array(0.0), tv], 1)
    lb, gb = _run(ops.SolveODEAdjointBatchBackward(solver), [Y0, P, fixed, np.stack([grads, grads]), np.array(0.0), tv], 2)
    np.testing.assert_array_equal(yb[0], y)
    np.testing.assert_array_equal(gb[0], grad)
    assert yb.shape == (2, 21, 2) and gb.shape == (2, 2) and lb.shape == (2, 2)
    # EvalRhs (d/dtvals wiring): rhs along the solution
    rhs, = _run(ops.EvalRhs(solver), [params, fixed, y, tv], 1)
    a, b, c, dl = 0.1, 0.2, 0.3, 0.4
    np.testing.assert_allclose(rhs[:, 0], a * y[:, 0] - b * y[:, 1] * y[:, 0], rtol=1e-13)
    np.testing.assert_allclose(rhs[:, 1], dl * y[:, 0] * y[:, 1] - c * y[:, 1], rtol=1e-13)


def test_forward_sensitivity_op_perform(ops):
    from sunode_amd.solver import Solver
    prob = make_problem("lv")
    solver = Solver(prob, abstol=1e-9, reltol=1e-9, sens_mode="simultaneous")
    y0 = np.array([1.0, 0.1]); params = np.array([0.1, 0.2]); fixed = np.array([0.3, 0.4])
    tv = np.linspace(0, 10, 21)
    op = ops.SolveODE(solver)
    assert op._sens0.shape == (2, 2) and not op._sens0.any()
    y, sens = _run(op, [y0, params, fixed, np.array(0.0), tv], 2)
    yd, sd, st, _ = solver.solve_sens_batch(0.0, tv, y0[None], params[None], fixed, np.zeros((2, 2)))
    assert st[0] == 0
    np.testing.assert_array_equal(y, yd[0])
    np.testing.assert_array_equal(sens, sd[0])
    assert sens.shape == (21, 2, 2) and np.abs(sens[-1]).max() > 0.1


def _lv_rhs(t, y, p):
    return {"hares": p.alpha * y.hares - p.beta * y.lynx * y.hares,
            "lynx": p.delta * y.hares * y.lynx - p.gamma * y.lynx}


def _oracle_gradients(tv, y0, params, grads, tol):
    """dL/dy0, dL/d(alpha, beta), y(t) for L = sum(grads * y) from the CPU oracle (forward + adjoint)."""
    from tests.helpers import make_oracle
    prob = make_problem("lv")
    orc = make_oracle("lv")
    cfg = orc.config(rtol=tol, atol=tol, rtolB=tol, atolB=tol, rtolQB=tol, atolQB=tol)
    y, st, _ = orc.solve_forward(cfg, y0[None], params[None, :2], params[None, 2:], 0.0, tv)
    g, lam, st2, _ = orc.solve_backward(cfg, tv[-1], 0.0, tv, grads)
    assert st[0] == 0 and st2[0] == 0
    return y[0], -lam[0], g[0]


def test_solve_ivp_graph_and_adjoint_grad_chain_vs_oracle(ops):
    """``solve_ivp`` builds the graph, ``SolveODEAdjoint.grad`` wires forward Op -> SolveODEAdjointBackward ->
    EvalRhs; evaluating that graph must give the ORACLE's states and gradients (bit for bit: same arithmetic)
    and d/dtvals = (rhs(y(t_i)) * g_i).sum(-1)  (reference wrappers/as_pytensor.py:294-308)."""
    pytensor = pytest.importorskip("pytensor")
    if not hasattr(pytensor, "evaluate"):
        pytest.skip("graph evaluation helper of the stub only")
    pt = importlib.import_module("pytensor.tensor")
    tol = 1e-9
    alpha, beta = pt.dscalar("alpha"), pt.dscalar("beta")
    hares0 = pt.dscalar("hares0")
    tv = np.linspace(0, 10, 21)
    sol, flat, problem, solver, y0_flat, ps_flat = ops.solve_ivp(
        t0=0.0, y0={"hares": (hares0, ()), "lynx": np.array(0.1)},
        params={"alpha": (alpha, ()), "beta": (beta, ()), "gamma": np.array(0.3), "delta": np.array(0.4)},
        tvals=tv, rhs=_lv_rhs, derivatives="adjoint",
        solver_kwargs=dict(abstol=tol, reltol=tol, backward_abstol=tol, backward_reltol=tol, quad_abstol=tol,
                           quad_reltol=tol))
    assert [".".join(p) for p in problem.params_subset.subset_paths] == ["alpha", "beta"]
    givens = {alpha: 0.1, beta: 0.2, hares0: 1.0}
    y = pytensor.evaluate(flat, givens)
    assert y.shape == (21, 2)
    np.testing.assert_array_equal(pytensor.evaluate(sol["lynx"], givens), y[:, 1])
    # cotangent of L = sum(w * y): hand it to the Op's grad() exactly as pytensor.grad would
    w = np.cos(np.arange(42.0)).reshape(21, 2)
    node = flat.owner
    assert type(node.op).__name__ == "SolveODEAdjoint"
    gl = node.op.grad(node.inputs, [pt.as_tensor_variable(w)])
    assert len(gl) == 5
    assert type(gl[2]).__name__ == "NotImplementedGrad" and type(gl[3]).__name__ == "NotImplementedGrad"
    d_y0, d_params, d_tvals = pytensor.evaluate([gl[0], gl[1], gl[4]], givens)
    yo, dy0_o, dp_o = _oracle_gradients(tv, np.array([1.0, 0.1]), np.array([0.1, 0.2, 0.3, 0.4]), w, tol)
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(d_params, dp_o)
    np.testing.assert_array_equal(d_y0, dy0_o)
    rhs = np.stack([0.1 * yo[:, 0] - 0.2 * yo[:, 1] * yo[:, 0], 0.4 * yo[:, 0] * yo[:, 1] - 0.3 * yo[:, 1]], axis=1)
    np.testing.assert_allclose(d_tvals, (rhs * w).sum(-1), rtol=1e-12)
    # finite-difference check of one entry through the graph itself (independent of the adjoint machinery)
    eps = 1e-6
    lp = (pytensor.evaluate(flat, {**givens, alpha: 0.1 + eps}) * w).sum()
    lm = (pytensor.evaluate(flat, {**givens, alpha: 0.1 - eps}) * w).sum()
    assert abs((lp - lm) / (2 * eps) - d_params[0]) < 2e-5 * max(1.0, abs(d_params[0]))


def test_solve_ivp_graph_of_a_model_with_the_reference_helper_functions(ops):
    """The normal PyMC model shape through the graph builder: log-parameterised rates, ``expit`` switch, ``logaddexp``
    threshold and a degree-4 ``interpolate_spline`` input whose coefficients are a pytensor VECTOR -- imported under the
    reference's names (``sunode.symode.lambdify``).  The evaluated graph equals the ORACLE's states and gradients bit
    for bit (the helper functions run as csrc/sa_math.h on both sides) and a finite difference through the graph."""
    pytensor = pytest.importorskip("pytensor")
    if not hasattr(pytensor, "evaluate"):
        pytest.skip("graph evaluation helper of the stub only")
    from tests.helpers import make_oracle
    from tools.problems import forcing, forcing_batch
    pt = importlib.import_module("pytensor.tensor")
    tol = 1e-8
    log_r, log_K, a, w = pt.dscalar("log_r"), pt.dscalar("log_K"), pt.dscalar("a"), pt.dvector("w")
    d = forcing_batch(1)
    tv = d["tvals"]
    sol, flat, problem, solver, y0_flat, ps_flat = ops.solve_ivp(
        t0=0.0, y0={"x": np.array(d["y0"][0, 0]), "z": np.array(d["y0"][0, 1]), "c": np.array(0.0)},
        params={"log_r": (log_r, ()), "log_K": (log_K, ()), "w": (w, (5,)), "k": np.array(2.0), "t_mid": np.array(3.0),
                "a": (a, ()), "s": np.array(1.5)},
        tvals=tv, rhs=forcing, derivatives="adjoint",
        solver_kwargs=dict(abstol=tol, reltol=tol, backward_abstol=tol, backward_reltol=tol, quad_abstol=tol,
                           quad_reltol=tol))
    assert [".".join(p) for p in problem.params_subset.subset_paths] == ["log_r", "log_K", "w", "a"]
    ps = d["ps"][0]
    givens = {log_r: ps[0], log_K: ps[1], w: ps[2:7], a: ps[7]}
    y = pytensor.evaluate(flat, givens)
    g_out = d["grads"][0]
    node = flat.owner
    gl = node.op.grad(node.inputs, [pt.as_tensor_variable(g_out)])
    d_y0, d_params = pytensor.evaluate([gl[0], gl[1]], givens)
    orc = make_oracle("forcing")
    cfg = orc.config(rtol=tol, atol=tol, rtolB=tol, atolB=tol, rtolQB=tol, atolQB=tol)
    yo, so, _ = orc.solve_forward(cfg, d["y0"], d["ps"], d["pr"], 0.0, tv)
    go, lo, sbo, _ = orc.solve_backward(cfg, tv[-1], 0.0, tv, g_out)
    assert so[0] == 0 and sbo[0] == 0
    np.testing.assert_array_equal(y, yo[0])
    np.testing.assert_array_equal(d_params, go[0])
    np.testing.assert_array_equal(d_y0, -lo[0])
    eps = 1e-6
    wp, wm = ps[2:7].copy(), ps[2:7].copy()
    wp[2] += eps; wm[2] -= eps
    lp = (pytensor.evaluate(flat, {**givens, w: wp}) * g_out).sum()
    lm = (pytensor.evaluate(flat, {**givens, w: wm}) * g_out).sum()
    assert abs((lp - lm) / (2 * eps) - d_params[4]) < 2e-5 * max(1.0, abs(d_params[4]))


def test_solve_ivp_forward_sensitivity_grad_chain(ops):
    """derivatives='forward': SolveODE.grad contracts the sensitivities; a cotangent on the sensitivity output
    is refused like in the reference (:253)."""
    pytensor = pytest.importorskip("pytensor")
    if not hasattr(pytensor, "evaluate"):
        pytest.skip("graph evaluation helper of the stub only")
    pt = importlib.import_module("pytensor.tensor")
    grad_mod = importlib.import_module("pytensor.gradient")
    tol = 1e-9
    alpha, beta = pt.dscalar("alpha"), pt.dscalar("beta")
    tv = np.linspace(0, 10, 21)
    out = ops.solve_ivp(t0=0.0, y0={"hares": np.array(1.0), "lynx": np.array(0.1)},
                        params={"alpha": (alpha, ()), "beta": (beta, ()), "gamma": np.array(0.3),
                                "delta": np.array(0.4)},
                        tvals=tv, rhs=_lv_rhs, derivatives="forward",
                        solver_kwargs=dict(abstol=tol, reltol=tol, sens_mode="simultaneous"))
    flat, flat_sens = out[1], out[6]
    givens = {alpha: 0.1, beta: 0.2}
    w = np.cos(np.arange(42.0)).reshape(21, 2)
    node = flat.owner
    gl = node.op.grad(node.inputs, [pt.as_tensor_variable(w), grad_mod.DisconnectedType()()])
    d_params = pytensor.evaluate(gl[1], givens)
    _, _, dp_o = _oracle_gradients(tv, np.array([1.0, 0.1]), np.array([0.1, 0.2, 0.3, 0.4]), w, tol)
    np.testing.assert_allclose(d_params, dp_o, rtol=2e-6)          # two different gradient methods at tol 1e-9
    assert pytensor.evaluate(flat_sens, givens).shape == (21, 2, 2)
    with pytest.raises(NotImplementedError):
        node.op.grad(node.inputs, [pt.as_tensor_variable(w), pt.as_tensor_variable(np.zeros((21, 2, 2)))])


def test_batched_op_grad_chain_vs_oracle(ops):
    """``SolveODEAdjointBatch.grad``: forward Op -> SolveODEAdjointBatchBackward -> EvalRhsBatch, evaluated through the
    graph for three draws with per-draw cotangents; every row must equal the ORACLE's gradients of that draw bit for
    bit, and d/dtvals (the grid is shared by the draws) = sum_b (rhs(y_b(t_i)) * g_bi).sum(-1)."""
    pytensor = pytest.importorskip("pytensor")
    if not hasattr(pytensor, "evaluate"):
        pytest.skip("graph evaluation helper of the stub only")
    pt = importlib.import_module("pytensor.tensor")
    from sunode_amd.solver import AdjointSolver
    tol = 1e-9
    solver = AdjointSolver(make_problem("lv"), abstol=tol, reltol=tol, backward_abstol=tol, backward_reltol=tol,
                           quad_abstol=tol, quad_reltol=tol)
    tv = np.linspace(0, 10, 21)
    Y0 = np.array([[1.0, 0.1], [1.1, 0.12], [0.9, 0.2]])
    P = np.array([[0.1, 0.2], [0.12, 0.18], [0.09, 0.25]])
    fixed = np.array([0.3, 0.4])
    W = np.cos(np.arange(3 * 42.0)).reshape(3, 21, 2)
    y0v, pv = pt.dmatrix("y0"), pt.dmatrix("params")
    flat = ops.SolveODEAdjointBatch(solver)(y0v, pv, fixed, 0.0, tv)
    node = flat.owner
    gl = node.op.grad(node.inputs, [pt.as_tensor_variable(W)])
    assert len(gl) == 5 and type(gl[2]).__name__ == "NotImplementedGrad" and type(gl[3]).__name__ == "NotImplementedGrad"
    givens = {y0v: Y0, pv: P}
    y, d_y0, d_params, d_tvals = pytensor.evaluate([flat, gl[0], gl[1], gl[4]], givens)
    want_tv = np.zeros(len(tv))
    for b in range(3):
        yo, dy0_o, dp_o = _oracle_gradients(tv, Y0[b], np.concatenate([P[b], fixed]), W[b], tol)
        np.testing.assert_array_equal(y[b], yo)
        np.testing.assert_array_equal(d_params[b], dp_o)
        np.testing.assert_array_equal(d_y0[b], dy0_o)
        rhs = np.stack([P[b, 0] * yo[:, 0] - P[b, 1] * yo[:, 1] * yo[:, 0],
                        0.4 * yo[:, 0] * yo[:, 1] - 0.3 * yo[:, 1]], axis=1)
        want_tv += (rhs * W[b]).sum(-1)
    np.testing.assert_allclose(d_tvals, want_tv, rtol=1e-12)
    # one entry of d/dtvals against a finite difference of the loss through the graph
    k, eps = 7, 1e-6
    def loss(tvk):
        tvp = tv.copy(); tvp[k] = tvk
        return (pytensor.evaluate(ops.SolveODEAdjointBatch(solver)(y0v, pv, fixed, 0.0, tvp), givens) * W).sum()
    fd = (loss(tv[k] + eps) - loss(tv[k] - eps)) / (2 * eps)
    assert abs(fd - d_tvals[k]) < 1e-5 * max(1.0, abs(d_tvals[k]))
