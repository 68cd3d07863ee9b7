"""The numeric halves (``perform``) of the pytensor Ops against the solvers they wrap.

pytensor is not installed in this image, so a stub package (tests/stubs/pytensor: just the names the wrapper
module touches) stands in for it; with the real pytensor installed the same test runs against it.  Graph
construction and ``grad`` wiring need the real package and are not covered here."""
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
