"""pytensor front-end: ``solve_ivp`` + Ops with the reference's names and wiring.

Counterpart of /root/reference/sunode/wrappers/as_pytensor.py (``solve_ivp`` ``:20-137``,
``EvalRhs`` ``:140-183``, ``SolveODEAdjoint`` ``:266-308``, ``SolveODEAdjointBackward``
``:311-344``) on top of the HIP engine.  Both gradient paths are wired: ``derivatives='adjoint'``
(``SolveODEAdjoint``) and ``derivatives='forward'`` (``SolveODE``, forward sensitivities, ``:186-264``).
On top of the reference's per-draw Ops there is a batched pair (``SolveODEAdjointBatch`` / ``...BatchBackward``) whose
leading axis is the parameter draw, which is what actually feeds a GPU.

pytensor is an optional dependency: importing this module without it raises ``ImportError``.
Gradient wiring (reference ``:294-308``): d/dy0 = -lamda, d/dparams = grad_out,
d/dtvals = (rhs(y(t_i)) * g_i).sum(-1); fixed parameters and t0 are not differentiable.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional

import numpy as np

try:
    import pytensor.tensor as pt
    from pytensor.gradient import DisconnectedType, grad_not_implemented
    from pytensor.graph.basic import Constant, Variable
    from pytensor.graph.op import Op
except ImportError as exc:  # pragma: no cover - depends on the environment
    raise ImportError("sunode_amd.wrappers.as_pytensor needs pytensor, which is not installed") from exc

from sunode_amd.dtypesubset import as_flattened
from sunode_amd.solver import AdjointSolver, Solver, SolverError, initial_sensitivities
from sunode_amd.symode.problem import SympyProblem


def _static_dims(vals: Any, label: Optional[str] = None) -> Any:
    """Replace every tensor leaf of a nested dict by its (static) shape / dim names."""
    if isinstance(vals, dict):
        return {key: _static_dims(item, key) for key, item in vals.items()}
    if isinstance(vals, tuple):
        tensor, dims = vals
    else:
        tensor, dims = vals, pt.as_tensor_variable(vals, dtype="float64").type.shape
        if any(d is None for d in dims):
            raise ValueError("Shapes of tensors need to be statically known or given explicitly.")
    if isinstance(dims, (str, int)):
        dims = (dims,)
    tensor = pt.as_tensor_variable(tensor, dtype="float64")
    if tensor.ndim != len(dims):
        raise ValueError(f"Dimension mismatch for {label}: Value has rank {tensor.ndim}, "
                         f"but {len(dims)} was specified.")
    return dims


def _concat_flat(flat: Dict[Any, Any], paths) -> Any:
    pieces = []
    for path in paths:
        item = flat[path]
        if isinstance(item, tuple):
            item = item[0]
        pieces.append(pt.as_tensor_variable(item, dtype="float64").reshape((-1,)))
    if not pieces:
        return pt.as_tensor_variable(np.zeros(0), dtype="float64")
    return pt.concatenate(pieces)


def solve_ivp(t0, y0, params, tvals, rhs: Callable, derivatives: str = "adjoint", coords=None,
              make_solver=None, derivative_subset=None, solver_kwargs=None, simplify=None):
    """Build the pytensor graph of an ODE solution (reference ``solve_ivp``).

    Returns ``(solution_dict, flat_solution, problem, solver, y0_flat, params_subs_flat)``."""
    solver_kwargs = dict(solver_kwargs or {})
    if derivatives not in ("adjoint", "forward"):
        raise ValueError("derivatives must be 'adjoint' or 'forward'")
    if derivatives == "forward" and "sens_mode" not in solver_kwargs:
        raise ValueError("When `derivatives='forward'`, the `solver_kwargs` must contain one of "
                         "`sens_mode={\"simultaneous\" | \"staggered\"}`.")

    y0_dims = _static_dims(y0)
    params_dims = _static_dims(params)
    flat_params = as_flattened(params)
    if derivative_subset is None:
        # differentiate w.r.t. every parameter that is a non-constant graph variable (reference :72-81)
        derivative_subset = []
        for path, item in flat_params.items():
            tensor = item[0] if isinstance(item, tuple) else item
            if isinstance(tensor, Variable) and not isinstance(tensor, Constant):
                derivative_subset.append(path)

    problem = SympyProblem(params_dims, y0_dims, rhs, derivative_subset, coords=coords, simplify=simplify)
    params_subs_flat = _concat_flat(flat_params, problem.params_subset.subset_paths)
    params_rem_flat = _concat_flat(flat_params, problem.params_subset.remainder.subset_paths)
    y0_flat = _concat_flat(as_flattened(y0), problem.state_subset.paths)
    t0 = pt.as_tensor_variable(t0, dtype="float64")
    tvals = pt.as_tensor_variable(tvals, dtype="float64")

    if derivatives == "forward":       # reference :127-134: one more pair of return values
        solver = make_solver(problem, **solver_kwargs) if make_solver else Solver(problem, **solver_kwargs)
        wrapper = SolveODE(solver)
        flat_solution, flat_sens = wrapper(y0_flat, params_subs_flat, params_rem_flat, t0, tvals)
        solution = problem.flat_solution_as_dict(flat_solution)
        return solution, flat_solution, problem, solver, y0_flat, params_subs_flat, flat_sens, wrapper
    solver = make_solver(problem, **solver_kwargs) if make_solver else AdjointSolver(problem, **solver_kwargs)
    flat_solution = SolveODEAdjoint(solver)(y0_flat, params_subs_flat, params_rem_flat, t0, tvals)
    solution = problem.flat_solution_as_dict(flat_solution)
    return solution, flat_solution, problem, solver, y0_flat, params_subs_flat


class EvalRhs(Op):
    """rhs(t_i, y_i) for every output time (needed only for d/dtvals)."""
    itypes = [pt.dvector, pt.dvector, pt.dmatrix, pt.dvector]      # params, params_fixed, y, tvals
    otypes = [pt.dmatrix]
    __props__ = ("_solver_id",)

    def __init__(self, solver):
        self._solver = solver
        self._solver_id = id(solver)

    def perform(self, node, inputs, outputs):
        params, params_fixed, y, tvals = inputs
        eng = self._solver._engine()
        n_t = len(tvals)
        fixed = self._solver._problem.extend_remainder(params_fixed) if len(params_fixed) else params_fixed
        res = eng.eval_callbacks(tvals, y, np.zeros_like(y), np.tile(params, (n_t, 1)), np.tile(fixed, (n_t, 1)))
        if res["codes"][:, 0].any():
            raise ValueError("Bad ode rhs return code: 1")
        outputs[0][0] = res["rhs"]


class EvalRhsBatch(Op):
    """rhs(t_i, y_bi) for every draw b and output time i (d/dtvals of the batched Op)."""
    itypes = [pt.dmatrix, pt.dvector, pt.dtensor3, pt.dvector]      # params [B,p], params_fixed [r], y [B,n_t,n], tvals
    otypes = [pt.dtensor3]
    __props__ = ("_solver_id",)

    def __init__(self, solver):
        self._solver = solver
        self._solver_id = id(solver)

    def perform(self, node, inputs, outputs):
        params, params_fixed, y, tvals = inputs
        eng = self._solver._engine()
        B, n_t, n = y.shape
        fixed = self._solver._problem.extend_remainder(params_fixed) if len(params_fixed) else params_fixed
        res = eng.eval_callbacks(np.tile(tvals, B), y.reshape(B * n_t, n), np.zeros((B * n_t, n)),
                                 np.repeat(params, n_t, axis=0), np.tile(fixed, (B * n_t, 1)))
        if res["codes"][:, 0].any():
            raise ValueError("Bad ode rhs return code: 1")
        outputs[0][0] = res["rhs"].reshape(B, n_t, n)


class SolveODE(Op):
    """Forward solve + forward sensitivities of one draw (reference ``SolveODE``, :186-264)."""
    itypes = [pt.dvector, pt.dvector, pt.dvector, pt.dscalar, pt.dvector]   # y0, params, fixed, t0, tvals
    otypes = [pt.dmatrix, pt.dtensor3]                                       # y_out, sens_out [n_t, p, n]
    __props__ = ("_solver_id",)

    def __init__(self, solver):
        self._solver = solver
        self._solver_id = id(solver)
        self._sens0 = initial_sensitivities(solver._problem)

    def perform(self, node, inputs, outputs):
        y0, params, params_fixed, t0, tvals = inputs
        y, sens, status, _ = self._solver.solve_sens_batch(float(t0), tvals, y0[None], params[None], params_fixed,
                                                           self._sens0)
        outputs[0][0] = y[0]            # failed solves are NaN-filled by the engine (reference :245-247)
        outputs[1][0] = sens[0]

    def grad(self, inputs, g):
        g, g_sens = g
        _, params, params_fixed, t0, tvals = inputs
        # as the reference (:253): a loss that depends on the sensitivity output would need second-order
        # sensitivities; fail loudly instead of returning an incomplete gradient
        if not isinstance(getattr(g_sens, "type", None), DisconnectedType):
            raise NotImplementedError("SolveODE: gradients through the sensitivity output are not implemented")
        solution, sens = self(*inputs)
        return [
            pt.zeros_like(inputs[0]),
            pt.sum(g[:, None, :] * sens, (0, -1)),
            grad_not_implemented(self, 2, params_fixed),
            grad_not_implemented(self, 3, t0),
            (EvalRhs(self._solver)(params, params_fixed, solution, tvals) * g).sum(-1),
        ]


class SolveODEAdjoint(Op):
    itypes = [pt.dvector, pt.dvector, pt.dvector, pt.dscalar, pt.dvector]   # y0, params, fixed, t0, tvals
    otypes = [pt.dmatrix]
    __props__ = ("_solver_id",)

    def __init__(self, solver):
        self._solver = solver
        self._solver_id = id(solver)

    def perform(self, node, inputs, outputs):
        y0, params, params_fixed, t0, tvals = inputs
        y, status, _ = self._solver.solve_forward_batch(float(t0), tvals, y0[None], params[None], params_fixed)
        outputs[0][0] = y[0]            # failed solves are NaN-filled by the engine (reference :289-290)

    def grad(self, inputs, g):
        g, = g
        y0, params, params_fixed, t0, tvals = inputs
        solution = self(*inputs)
        lamda, gradient = SolveODEAdjointBackward(self._solver)(y0, params, params_fixed, g, t0, tvals)
        return [
            -lamda,
            gradient,
            grad_not_implemented(self, 2, params_fixed),
            grad_not_implemented(self, 3, t0),
            (EvalRhs(self._solver)(params, params_fixed, solution, tvals) * g).sum(-1),
        ]


class SolveODEAdjointBackward(Op):
    itypes = [pt.dvector, pt.dvector, pt.dvector, pt.dmatrix, pt.dscalar, pt.dvector]
    otypes = [pt.dvector, pt.dvector]                                 # lamda, gradient
    __props__ = ("_solver_id",)

    def __init__(self, solver):
        self._solver = solver
        self._solver_id = id(solver)

    def perform(self, node, inputs, outputs):
        y0, params, params_fixed, grads, t0, tvals = inputs
        # like the reference (:332-336) the forward pass is repeated: the trajectory arena belongs to it
        self._solver.solve_forward_batch(float(t0), tvals, y0[None], params[None], params_fixed)
        grad_out, lamda_out, status, _ = self._solver.solve_backward_batch(
            float(tvals[-1]), float(t0), tvals, np.ascontiguousarray(grads)[None])
        outputs[0][0] = lamda_out[0]
        outputs[1][0] = grad_out[0]


class SolveODEAdjointBatch(Op):
    """Batched forward solve: y0 [B,n], params [B,p], params_fixed [r] (shared), t0, tvals -> [B,n_t,n]."""
    itypes = [pt.dmatrix, pt.dmatrix, pt.dvector, pt.dscalar, pt.dvector]
    otypes = [pt.dtensor3]
    __props__ = ("_solver_id",)

    def __init__(self, solver):
        self._solver = solver
        self._solver_id = id(solver)

    def perform(self, node, inputs, outputs):
        y0, params, params_fixed, t0, tvals = inputs
        y, _, _ = self._solver.solve_forward_batch(float(t0), tvals, y0, params, params_fixed)
        outputs[0][0] = y

    def grad(self, inputs, g):
        g, = g
        y0, params, params_fixed, t0, tvals = inputs
        solution = self(*inputs)
        lamda, gradient = SolveODEAdjointBatchBackward(self._solver)(y0, params, params_fixed, g, t0, tvals)
        # the output grid is shared by the draws: its gradient sums over the batch axis (per draw: reference :294-308)
        d_tvals = (EvalRhsBatch(self._solver)(params, params_fixed, solution, tvals) * g).sum(-1).sum(0)
        return [-lamda, gradient, grad_not_implemented(self, 2, params_fixed),
                grad_not_implemented(self, 3, t0), d_tvals]


class SolveODEAdjointBatchBackward(Op):
    itypes = [pt.dmatrix, pt.dmatrix, pt.dvector, pt.dtensor3, pt.dscalar, pt.dvector]
    otypes = [pt.dmatrix, pt.dmatrix]
    __props__ = ("_solver_id",)

    def __init__(self, solver):
        self._solver = solver
        self._solver_id = id(solver)

    def perform(self, node, inputs, outputs):
        y0, params, params_fixed, grads, t0, tvals = inputs
        self._solver.solve_forward_batch(float(t0), tvals, y0, params, params_fixed)
        grad_out, lamda_out, _, _ = self._solver.solve_backward_batch(
            float(tvals[-1]), float(t0), tvals, np.ascontiguousarray(grads))
        outputs[0][0] = lamda_out
        outputs[1][0] = grad_out
