"""``SympyProblem``: symbolic definition of an ODE and of everything the adjoint needs.

Keeps the constructor, attributes and semantics of the reference class
(/root/reference/sunode/symode/problem.py:24-158) -- same symbols
(states ``positive=True`` ``:78``, parameters ``real=True`` ``:79``, ``time``
``:67``), same derived objects (``J = df/dy`` ``:142``, ``df/dp`` ``:144``,
adjoint rhs ``-lam^T J`` ``:147``, quadrature rhs ``lam^T df/dp`` ``:148``) and
the same ``user_data_dtype`` (``:150-158``) -- but instead of numba C callbacks
(``:251-433``) it emits HIP C++ device functions (see ``codegen.py``) that the
batched BDF kernels include.  The ``make_*`` methods still exist and return
host callables with the reference's call signatures (used by ``EvalRhs`` and by
user code that evaluates the right-hand side directly); they are host-side
conveniences, not part of the device path.
"""
from __future__ import annotations

import os

from itertools import product
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np
import pandas as pd
import sympy as sym

from sunode_amd import dtypesubset
from sunode_amd.symode import codegen

Path = Tuple[str, ...]
Shape = Tuple[int, ...]
#: fixed-parameter sub-expressions with at least this many operations are evaluated once on the host
HOIST_MIN_OPS = 8
#: a callback touching at least this many fixed parameters in a scattered order gets an access-order copy
PACK_MIN_SYMBOLS = 256
#: a callback whose outputs contain a dense (fixed parameter) x (state | adjoint state) product block with at least
#: this many entries gets it emitted as ONE lane-parallel matrix-vector product (codegen.SA_MATVEC)
MATVEC_MIN_ENTRIES = 256
MATVEC_MIN_FILL = 0.5

data_dtype = np.dtype(np.float64)


def _identity(expr):
    return expr


def _scalarize(item: np.ndarray) -> Any:
    if hasattr(item, "shape") and item.shape == ():
        return item.item()
    return item


def extract_matvec(exprs, vsyms, fixed, tag):
    """Split the outputs ``e_i`` of a vector callback as ``e_i = sum_j M_ij v_j + S + r_i``:

    * ``M_ij = +-(one fixed-parameter symbol)``: a dense constant block -- becomes one matrix-vector product
      evaluated lane-parallel by the kernels (``SA_MATVEC``) instead of ``n_out * n_in`` scalar statements;
    * ``S = sum_j v_j u_j``: the part of the remaining coefficient of ``v_j`` that is the same for (nearly) all
      outputs (the rank-one ``1 (u . v)`` term a dense Jacobian transpose produces), computed once;
    * ``r_i``: whatever is left, as ordinary straight-line code.

    ``v`` is the state vector (right-hand side) or the adjoint state (adjoint right-hand side).  Pure term
    re-grouping: the value of every output is unchanged up to rounding.  Returns None when there is no block worth
    it, else ``(new_exprs, mv_symbols, entries)`` with ``entries[j][i] = (sign, fixed symbol) | None`` and
    ``mv_symbols[i]`` standing for the i-th product in ``new_exprs``."""
    n_out = len(exprs)
    vidx = {sy: j for j, sy in enumerate(vsyms)}
    vset = set(vsyms)
    n_in = len(vsyms)
    if n_out * n_in < MATVEC_MIN_ENTRIES:
        return None
    M: Dict[Tuple[int, int], Tuple[int, Any]] = {}
    rest: List[Dict[int, List[Any]]] = [dict() for _ in range(n_out)]
    other: List[List[Any]] = [[] for _ in range(n_out)]
    for i, e in enumerate(exprs):
        for term in sym.Add.make_args(sym.sympify(e)):
            c, factors = term.as_coeff_mul()
            vf = [f for f in factors if f in vset]
            others = [f for f in factors if f not in vset]
            if len(vf) != 1 or any(f.free_symbols & vset for f in others):
                other[i].append(term)
                continue
            j = vidx[vf[0]]
            coef = c * sym.Mul(*others)            # a numeric factor distributes over a sum here
            for a in sym.Add.make_args(coef):
                ca, fa = a.as_coeff_mul()
                if len(fa) == 1 and fa[0] in fixed and (ca == 1 or ca == -1) and (i, j) not in M:
                    M[(i, j)] = (int(ca), fa[0])
                else:
                    rest[i].setdefault(j, []).append(a)
    # one canonical term list per (output, v_j): an unexpanded right-hand side such as y0*(a + b) + a*y0 leaves the same
    # atom twice in rest[i][j]; merged here (2*a) so that the shared-term bookkeeping below removes and adds whole terms
    for i in range(n_out):
        for j in list(rest[i]):
            terms = [a for a in sym.Add.make_args(sym.Add(*rest[i][j])) if a != 0]
            if terms:
                rest[i][j] = terms
            else:
                del rest[i][j]
    cols = sorted({j for (_, j) in M})
    if not cols or len(M) < MATVEC_MIN_ENTRIES or len(M) < MATVEC_MIN_FILL * n_out * len(cols):
        return None
    # S: coefficient terms of v_j shared by (nearly) all outputs
    shared_terms = []
    for j in range(n_in):
        count: Dict[Any, int] = {}
        for i in range(n_out):
            for a in rest[i].get(j, ()):
                count[a] = count.get(a, 0) + 1
        uj = [a for a, k in count.items() if k >= 0.75 * n_out and k >= 4]
        if not uj:
            continue
        ujset = set(uj)
        for i in range(n_out):
            have = rest[i].get(j, [])
            kept = [a for a in have if a not in ujset]
            missing = [a for a in uj if a not in have]
            kept += [-a for a in missing]
            if kept:
                rest[i][j] = kept
            else:
                rest[i].pop(j, None)
        shared_terms.append(vsyms[j] * sym.Add(*uj))
    S = sym.Add(*shared_terms) if shared_terms else sym.Integer(0)
    mv = [sym.Symbol("sa_mv%s_%d" % (tag, i), real=True) for i in range(n_out)]
    new = []
    for i in range(n_out):
        pieces = [mv[i], S] + [vsyms[j] * sym.Add(*terms) for j, terms in sorted(rest[i].items())] + other[i]
        new.append(sym.Add(*pieces))
    entries = [[M.get((i, j)) for i in range(n_out)] for j in range(n_in)]
    return np.array(new, dtype=object), mv, entries


def extract_matfill(mat, fixed, tag):
    """Split the entries of a matrix callback (n x n, entry (i, j) stored column-major at slot j*n + i) as
    ``e_ij = M_ij + u_k + x_ij`` with ``M_ij = +-(one fixed-parameter symbol)`` (a constant block, zero where the
    entry has none), ``u_k`` a vector indexed by the row (axis 0) or the column (axis 1) -- the part shared by a
    whole line of the matrix, e.g. the rank-one term of a dense Jacobian -- and ``x_ij`` non-zero for few entries
    only (the diagonal).  The kernels then fill the matrix lane-parallel (``SA_MATFILL``: constant block + line
    vector) and evaluate straight-line code only for the exceptions, instead of n*n scalar statements.
    Returns None if there is no such structure, else ``(axis, u [n], exceptions {(i, j): x_ij}, entries)`` with
    ``entries[slot] = (sign, symbol) | None``."""
    from collections import Counter
    n = mat.shape[0]
    if n * n < MATVEC_MIN_ENTRIES or mat.shape != (n, n):
        return None
    m: Dict[Tuple[int, int], Tuple[int, Any]] = {}
    rest: Dict[Tuple[int, int], Any] = {}
    for i in range(n):
        for j in range(n):
            others = []
            for a in sym.Add.make_args(sym.sympify(mat[i, j])):
                ca, fa = a.as_coeff_mul()
                if (i, j) not in m and len(fa) == 1 and fa[0] in fixed and (ca == 1 or ca == -1):
                    m[(i, j)] = (int(ca), fa[0])
                else:
                    others.append(a)
            rest[(i, j)] = sym.Add(*others)
    if len(m) < MATVEC_MIN_FILL * n * n:
        return None
    best = None
    for axis in (0, 1):
        u, cover = [], 0
        for k in range(n):
            line = [rest[(k, q)] if axis == 0 else rest[(q, k)] for q in range(n)]
            expr, cnt = Counter(line).most_common(1)[0]
            u.append(expr)
            cover += cnt
        if best is None or cover > best[0]:
            best = (cover, axis, u)
    cover, axis, u = best
    if cover < 0.75 * n * n:
        return None
    exceptions = {}
    for i in range(n):
        for j in range(n):
            k = i if axis == 0 else j
            if rest[(i, j)] != u[k]:
                exceptions[(i, j)] = rest[(i, j)] - u[k]
    entries = [m.get((slot % n, slot // n)) for slot in range(n * n)]
    return axis, u, exceptions, entries


class SympyProblem:
    def __init__(
        self,
        params: Dict[str, Any],
        states: Dict[str, Any],
        rhs_sympy: Callable[[sym.Symbol, Any, Any], Dict[str, Any]],
        derivative_params: List[Path],
        coords: Optional[Dict[str, pd.Index]] = None,
        simplify: Optional[Callable[[sym.Expr], sym.Expr]] = None,
    ):
        self.params_subset = dtypesubset.DTypeSubset(
            params, derivative_params, fixed_dtype=data_dtype, coords=coords)
        self.coords = self.params_subset.coords
        self.params_dtype = self.params_subset.dtype
        self.state_subset = dtypesubset.DTypeSubset(
            states, [], fixed_dtype=data_dtype, coords=self.coords)
        self.state_dtype = self.state_subset.dtype
        self._rhs_sympy_func = rhs_sympy
        self._simplify_func = simplify if simplify is not None else _identity
        self._simplify = np.vectorize(self._simplify_func, otypes=[object])

        self._sym_time = sym.Symbol("time", real=True)

        # -- symbols: one sympy symbol per scalar slot, named '<path>_<idx>' ----
        # (reference :70-95; scalar leaves get sympy's trailing underscore name)
        self._varmap: Dict[str, Tuple[Any, ...]] = {}
        self._c_slots: Dict[str, str] = {"time": "t"}

        state_syms = self._declare(self.state_subset, "state", positive=True)
        param_syms = self._declare(self.params_subset, "params", real=True)
        self._state_syms, self._param_syms = state_syms, param_syms

        sub_paths = set(self.params_subset.subset_paths)
        deriv = [param_syms[p].ravel() for p in self.params_subset.paths if p in sub_paths]
        fixed = [param_syms[p].ravel() for p in self.params_subset.paths if p not in sub_paths]
        self._sym_deriv_paramsvec = np.concatenate(deriv) if deriv else np.zeros((0,), dtype=object)
        self._sym_fixed_paramsvec = np.concatenate(fixed) if fixed else np.zeros((0,), dtype=object)
        svec = [state_syms[p].ravel() for p in self.state_subset.paths]
        self._sym_statevec = np.concatenate(svec) if svec else np.zeros((0,), dtype=object)

        for j, s in enumerate(self._sym_deriv_paramsvec):
            self._c_slots[s.name] = "SA_PS(%d)" % j
        for j, s in enumerate(self._sym_fixed_paramsvec):
            self._c_slots[s.name] = "SA_PR(%d)" % j
        for i, s in enumerate(self._sym_statevec):
            self._c_slots[s.name] = "SA_Y(%d)" % i

        self._sym_params = self.params_subset.as_dataclass(
            "Params", self._sym_deriv_paramsvec, self._sym_fixed_paramsvec, item_map=_scalarize)
        self._sym_states = self.state_subset.as_dataclass(
            "State", [], self._sym_statevec, item_map=_scalarize)

        dydt = self._make_dydt()
        self._sym_dydt = np.array(dydt).ravel()

        n, p = self.n_states, self.n_params
        self._sym_sens = sym.symarray("sens", (p, n))
        self._sym_lamda = sym.symarray("lamda", n)
        for i in range(n):
            self._varmap[self._sym_lamda[i].name] = ("lamda", (i,))
            self._c_slots[self._sym_lamda[i].name] = "SA_LAM(%d)" % i
        for idx in product(range(p), range(n)):
            self._varmap[self._sym_sens[idx].name] = ("sens", idx)

        if n:
            self._sym_dydt_jac = np.array(dydt.jacobian(list(self._sym_statevec))).reshape(n, n)
        else:
            self._sym_dydt_jac = np.zeros((0, 0), dtype=object)
        if n and p:
            self._sym_dydp = np.array(dydt.jacobian(list(self._sym_deriv_paramsvec))).reshape(n, p)
        else:
            self._sym_dydp = np.zeros((n, p), dtype=object)
        self._sym_dlamdadt = -self._sym_lamda @ self._sym_dydt_jac if n else np.zeros((0,), object)
        self._sym_quad_rhs = self._sym_lamda @ self._sym_dydp if (n and p) else np.zeros((p,), object)

        self.user_data_dtype = np.dtype([
            ("params", self.params_subset.dtype),
            ("tmp_nstates_nstates", np.float64, (n, n)),
            ("tmp_nparams_nstates", np.float64, (p, n)),
            ("tmp2_nparams_nstates", np.float64, (p, n)),
            ("error_states", self.state_dtype),
            ("error_rhs", np.float64, (n,)),
            ("error_jac", np.float64, (n, n)),
        ])
        self._native_source: Optional[str] = None
        self._native_cache = None
        self._hoisted: List[Any] = []
        self._packed: List[int] = []
        self._matvec: Dict[str, Dict[str, Any]] = {}
        self._matfill: Dict[str, Dict[str, Any]] = {}
        self._mv_index: List[int] = []
        self._mv_sign: List[float] = []
        self._hoist_fn = None
        self._host_funcs: Dict[str, Any] = {}

    # ------------------------------------------------------------------
    def __getstate__(self):
        # dynamic dataclasses / lambdified host functions are rebuilt on demand, not pickled
        state = dict(self.__dict__)
        for key in ("_sym_params", "_sym_states", "_host_funcs", "_simplify", "_hoist_fn"):
            state.pop(key, None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._host_funcs = {}
        self._hoist_fn = None
        self._simplify = np.vectorize(self._simplify_func, otypes=[object])

    def _declare(self, subset: dtypesubset.DTypeSubset, kind: str, **assumptions) -> Dict[Path, np.ndarray]:
        out: Dict[Path, np.ndarray] = {}
        for path, shape in subset.flat_shapes.items():
            arr = sym.symarray("_".join(path), shape, **assumptions)
            out[path] = arr
            for idx in product(*[range(k) for k in shape]):
                var = arr[idx]
                self._varmap[var.name] = (kind, *path) if idx == () else (kind, *path, idx)
        return out

    @property
    def n_states(self) -> int:
        return self.state_subset.n_items

    @property
    def n_params(self) -> int:
        """Number of *differentiated* parameters (reference problem.py:96-98)."""
        return self.params_subset.n_subset

    @property
    def n_remainder(self) -> int:
        return self.params_subset.n_items - self.params_subset.n_subset

    # ------------------------------------------------------------------
    def _make_dydt(self) -> sym.Matrix:
        """Call the user's rhs and flatten its nested result in state order
        (reference :160-230)."""
        rhs = self._rhs_sympy_func(self._sym_time, self._sym_states, self._sym_params)
        flat_dims = {k: names for k, (_, names) in dtypesubset.as_flattened(self.state_subset.dims).items()}

        def flatten(label: str, value: Any, shape: Shape, dims: Tuple[str, ...]) -> List[Any]:
            if hasattr(value, "shape"):
                if tuple(value.shape) != tuple(shape):
                    raise ValueError(
                        "Invalid shape for right-hand-side state %s. It is %s but we expected %s."
                        % (label, value.shape, shape))
                if isinstance(value, sym.NDimArray):
                    return list(value.reshape(int(np.prod(shape, dtype=int))))
                data = getattr(value, "data", value)      # xarray.DataArray duck-typing
                return list(np.asarray(data, dtype=object).reshape(-1))
            if isinstance(value, (list, tuple)):
                if not shape or len(value) != shape[0]:
                    raise ValueError("Invalid shape for right-hand-side state %s." % label)
                out: List[Any] = []
                for item in value:
                    out.extend(flatten(label, item, shape[1:], dims[1:]))
                return out
            if isinstance(value, dict):
                if not shape or len(value) != shape[0]:
                    raise ValueError("Invalid shape for right-hand-side state %s." % label)
                out = []
                for key in self.coords[dims[0]]:
                    out.extend(flatten(label, value[key], shape[1:], dims[1:]))
                return out
            if shape == ():
                return [value]
            raise ValueError("Unknown righ-hand-side for state %s." % label)

        def copy_nested(node: Any) -> Any:
            return {k: copy_nested(v) for k, v in node.items()} if isinstance(node, dict) else node

        rest = copy_nested(rhs)
        items: List[Any] = []
        for path in self.state_subset.paths:
            node = rest
            for key in path[:-1]:
                if key not in node:
                    raise ValueError("No right-hand-side for state %s" % ".".join(path))
                node = node[key]
            if path[-1] not in node:
                raise ValueError("No right-hand-side for state %s" % ".".join(path))
            value = node.pop(path[-1])
            items.extend(flatten(".".join(path), value,
                                 self.state_subset.flat_shapes[path], tuple(flat_dims[path])))
        leftover = dtypesubset.as_flattened(rest)
        if leftover:
            raise ValueError("Unknown state variables: %s" % [".".join(p) for p in leftover])
        return sym.Matrix(items) if items else sym.Matrix(0, 1, [])

    # ------------------------------------------------------------------
    # parameter plumbing on the host-side user_data record (reference :232-249)
    def make_user_data(self) -> np.ndarray:
        return np.zeros((), dtype=self.user_data_dtype).view(np.recarray)

    def update_params(self, user_data: np.ndarray, params: np.ndarray) -> None:
        user_data.params.fill(params)

    def update_subset_params(self, user_data: np.ndarray, params: np.ndarray) -> None:
        user_data.params.view(self.params_subset.subset_view_dtype).fill(params)

    def update_remaining_params(self, user_data: np.ndarray, params: np.ndarray) -> None:
        user_data.params.view(self.params_subset.remainder.subset_view_dtype).fill(params)

    def extract_params(self, user_data: np.ndarray, out: Optional[np.ndarray] = None) -> np.ndarray:
        if out is None:
            out = np.full((1,), np.nan, dtype=self.params_dtype)[0]
        out.fill(user_data.params)
        return out

    def flat_params(self, user_data: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """(ps, pr): differentiated / remaining parameters as flat float64 vectors."""
        full = np.array(user_data.params).reshape(1).view(np.float64) if self.params_dtype.itemsize \
            else np.zeros(0)
        return (np.ascontiguousarray(full[self.params_subset.subset_index]),
                np.ascontiguousarray(full[self.params_subset.remainder_index]))

    def flat_solution_as_dict(self, solution: np.ndarray) -> Dict[str, Any]:
        views = {}
        for path in self.state_subset.paths:
            shape = (-1,) + self.state_subset.flat_shapes[path]
            views[path] = solution[:, self.state_subset.flat_slices[path]].reshape(shape)
        return dtypesubset.as_nested(views)

    def solution_variables(self, tvals, solution, user_data, *, unstack_state=True, unstack_params=True):
        """The labelled arrays of a solution: ``{name: (dim names, values)}`` -- ``time``, then one entry per
        state leaf (``solution_<path>``, leading ``time`` axis) and one per parameter leaf
        (``parameters_<path>``), or the packed record arrays ``solution`` / ``parameters`` when not
        unstacked.  Built from the flat leaf tables (``flat_slices`` / ``flat_shapes`` / dim names); this is the
        content of the reference's ``solution_to_xarray`` (problem.py:100-145) without the xarray dependency."""
        tvals = np.asarray(tvals)
        solution = np.ascontiguousarray(solution, dtype=np.float64).reshape(len(tvals), self.n_states)
        flat_params = np.array(self.extract_params(user_data)).reshape(1).view(np.float64) \
            if self.params_dtype.itemsize else np.zeros(0)
        out: Dict[str, Tuple[Tuple[str, ...], np.ndarray]] = {"time": (("time",), tvals)}

        def add(name, dims, values):
            if name in out or name in self.coords:
                raise ValueError(f"Variable {name} is not unique.")
            out[name] = (tuple(dims), values)

        def leaves(subset, prefix, flat, lead_dims):
            for leaf in subset._leaves:
                block = flat[..., leaf.start:leaf.start + leaf.size]
                add("_".join((prefix,) + leaf.path), lead_dims + leaf.dim_names,
                    block.reshape(block.shape[:-1] + leaf.shape))

        if unstack_state:
            leaves(self.state_subset, "solution", solution, ("time",))
        else:
            add("solution", ("time",), solution.view(self.state_dtype)[..., 0])
        if unstack_params:
            leaves(self.params_subset, "parameters", flat_params, ())
        else:
            add("parameters", (), self.extract_params(user_data))
        return out

    def solution_to_xarray(self, tvals, solution, user_data, sensitivity=None,
                           *, unstack_state=True, unstack_params=True):
        """``xarray.Dataset`` of a solution (reference problem.py:100-145; xarray is optional here)."""
        if sensitivity is not None:
            raise NotImplementedError("sensitivities are not converted (neither does the reference, problem.py:106)")
        import xarray as xr
        data = xr.Dataset(coords=self.coords)
        for name, (dims, values) in self.solution_variables(
                tvals, solution, user_data, unstack_state=unstack_state, unstack_params=unstack_params).items():
            data[name] = (dims, values) if dims else values
        return data

    # ------------------------------------------------------------------
    # device / host native source
    def _native_exprs(self):
        """The four expression arrays the native callbacks are generated from, with expensive
        sub-expressions of the *fixed* (non-differentiated) parameters replaced by extra slots of
        the remainder vector: such a sub-expression has the same value in every callback
        evaluation of a solve (a rate-matrix column sum, say), so the host evaluates it once
        (``extend_remainder``) and the kernels -- and the CPU oracle, which compiles the same
        source -- just read it.  Problems without such sub-expressions are unaffected."""
        if self._native_cache is None:
            arrays = [self._simplify(np.array(a, dtype=object)) for a in
                      (self._sym_dydt, self._sym_dydt_jac, self._sym_dlamdadt, self._sym_quad_rhs,
                       np.array(self._sym_dydp, dtype=object).T)]
            fixed = set(self._sym_fixed_paramsvec)
            table: Dict[Any, sym.Symbol] = {}

            def hoistable(e):
                fs = e.free_symbols
                return bool(fs) and fs <= fixed and e.count_ops() >= HOIST_MIN_OPS

            def visit(e):
                if not isinstance(e, sym.Basic) or e.is_Atom:
                    return e
                if hoistable(e):
                    if e not in table:
                        table[e] = sym.Symbol("fixedpre_%d" % len(table), real=True)
                    return table[e]
                if not (e.free_symbols & fixed):
                    return e
                if e.is_Add or e.is_Mul:
                    # the fixed-parameter-only operands of a sum / product form one candidate
                    own = [a for a in e.args if a.free_symbols and a.free_symbols <= fixed]
                    if len(own) >= 2:
                        group = e.func(*own)
                        if hoistable(group):
                            rest = [a for a in e.args if not (a.free_symbols and a.free_symbols <= fixed)]
                            return e.func(visit(group), *[visit(a) for a in rest])
                return e.func(*[visit(a) for a in e.args])

            if len(fixed):
                arrays = [np.array([visit(sym.sympify(x)) for x in a.ravel()], dtype=object).reshape(a.shape)
                          for a in arrays]
            self._hoisted = list(table.keys())
            slots = dict(self._c_slots)
            for k, symbol in enumerate(table.values()):
                slots[symbol.name] = "SA_PR(%d)" % (self.n_remainder + k)

            # Dense (fixed parameter) x (state | adjoint state) blocks -> one matrix-vector product per callback,
            # reading a j-major, sign-folded copy of the block appended to the remainder vector (coalesced loads:
            # consecutive lanes own consecutive outputs).  Block k of callback `tag` starts at remainder slot
            # self._matvec[tag]["offset"]; entry (j, i) is sign * fixed parameter `index` (index -1: structural zero).
            self._matvec: Dict[str, Dict[str, Any]] = {}
            self._mv_index: List[int] = []
            self._mv_sign: List[float] = []
            fixed_or_hoisted = set(self._sym_fixed_paramsvec) | set(table.values())
            index_of_fixed = {sy: k for k, sy in enumerate(self._sym_fixed_paramsvec)}
            index_of_fixed.update({sy: self.n_remainder + k for k, sy in enumerate(table.values())})
            arrays = list(arrays)
            for pos, tag, vsyms in ((0, "f", list(self._sym_statevec)), (2, "a", list(self._sym_lamda))):
                if not len(fixed) or os.environ.get("SA_NO_MATVEC"):
                    continue
                found = extract_matvec(list(arrays[pos].ravel()), vsyms, fixed_or_hoisted, tag)
                if found is None:
                    continue
                new, mv, entries = found
                arrays[pos] = new.reshape(arrays[pos].shape)
                self._matvec[tag] = dict(n_out=len(mv), n_in=len(entries), offset=None, start=len(self._mv_index),
                                         vec="SA_Y" if tag == "f" else "SA_LAM")
                for col in entries:
                    for ent in col:
                        self._mv_index.append(-1 if ent is None else index_of_fixed[ent[1]])
                        self._mv_sign.append(0.0 if ent is None else float(ent[0]))
                for i, symbol in enumerate(mv):
                    slots[symbol.name] = "SA_MV(%s, %d)" % (tag, i)
            # Matrix callbacks (Jacobian, adjoint Jacobian) = constant block + line vector + few exceptions
            self._matfill: Dict[str, Dict[str, Any]] = {}
            n_st = self.n_states
            if len(fixed) and n_st and not os.environ.get("SA_NO_MATVEC"):
                jac_m = np.array(arrays[1], dtype=object).reshape(n_st, n_st)
                adj_m = np.array([[-jac_m[j, i] for j in range(n_st)] for i in range(n_st)], dtype=object)
                for tag, mat in (("j", jac_m), ("b", adj_m)):
                    found = extract_matfill(mat, fixed_or_hoisted, tag)
                    if found is None:
                        continue
                    axis, u, exceptions, entries = found
                    self._matfill[tag] = dict(n=n_st, axis=axis, u=u, exceptions=exceptions, offset=None,
                                              start=len(self._mv_index))
                    for ent in entries:
                        self._mv_index.append(-1 if ent is None else index_of_fixed[ent[1]])
                        self._mv_sign.append(0.0 if ent is None else float(ent[0]))

            # Access-order copies: a callback that walks a big block of fixed parameters with a large
            # stride (the adjoint right-hand side reads the rate matrix by columns) gets its own copy
            # of those parameters in the order it uses them, appended to the remainder vector by
            # ``extend_remainder``: sequential, cache-line friendly loads instead of one line per value.
            base = self.n_remainder + len(self._hoisted)
            index_of = {sy: k for k, sy in enumerate(self._sym_fixed_paramsvec)}
            self._packed: List[int] = []
            packed_arrays = []
            for tag, a in zip("fjaqs", arrays):
                order: List[Any] = []
                seen = set()
                pairs = adjacent = 0
                for x in a.ravel():
                    used = sorted((sy for sy in sym.sympify(x).free_symbols if sy in index_of),
                                  key=lambda sy: sy.name)
                    for u, v in zip(used, used[1:]):
                        pairs += 1
                        adjacent += abs(index_of[u] - index_of[v]) == 1
                    for sy in used:
                        if sy not in seen:
                            seen.add(sy)
                            order.append(sy)
                if len(order) >= PACK_MIN_SYMBOLS and pairs and adjacent < 0.5 * pairs:
                    sub = {}
                    for sy in order:
                        new = sym.Symbol("fixedpk%s_%07d" % (tag, len(self._packed)), real=True)
                        slots[new.name] = "SA_PR(%d)" % (base + len(self._packed))
                        self._packed.append(index_of[sy])
                        sub[sy] = new
                    a = np.array([sym.sympify(x).xreplace(sub) for x in a.ravel()], dtype=object).reshape(a.shape)
                packed_arrays.append(a)
            mv_base = base + len(self._packed)
            for info in list(self._matvec.values()) + list(self._matfill.values()):
                info["offset"] = mv_base + info.pop("start")
            self._native_cache = (packed_arrays, slots)
        return self._native_cache

    def _leaf_axes(self) -> Dict[str, Tuple[str, int, Tuple[Tuple[int, int, int], ...]]]:
        """symbol name -> (array macro, flat slot, ((stride, coordinate, axis length), ...)) for every state, adjoint
        state and user parameter symbol: where the symbol sits inside its leaf array.  The code generator uses it to
        recognise GROUP structure (codegen.find_lane_families): a model written as loops over M groups is equivariant
        under a relabelling of the groups, which acts on all axes of length M of all leaves at once."""
        out: Dict[str, Tuple[str, int, Tuple[Tuple[int, int, int], ...]]] = {}

        def add(symbols_by_path, subset, arr_of_slot):
            for path, arr in symbols_by_path.items():
                shape = tuple(subset.flat_shapes[path])
                strides = [int(np.prod(shape[k + 1:], dtype=int)) for k in range(len(shape))]
                for idx in product(*[range(k) for k in shape]):
                    name = arr[idx].name
                    text = self._c_slots.get(name)
                    m = codegen._LEAF_RE.match(text) if text else None
                    if m:
                        out[name] = (m.group(1), int(m.group(2)),
                                     tuple((strides[k], idx[k], shape[k]) for k in range(len(shape))))
        add(self._state_syms, self.state_subset, None)
        add(self._param_syms, self.params_subset, None)
        for name, info in list(out.items()):
            if info[0] == "SA_Y":           # the adjoint state mirrors the state's layout
                lam = self._sym_lamda[info[1]].name
                out[lam] = ("SA_LAM", info[1], info[2])
        return out

    @property
    def n_remainder_native(self) -> int:
        """Length of the remainder vector the native code reads (user part + hoisted values)."""
        self._native_exprs()
        return self.n_remainder + len(self._hoisted) + len(self._packed) + len(self._mv_index)

    def extend_remainder(self, pr: np.ndarray) -> np.ndarray:
        """[..., n_remainder] -> [..., n_remainder_native]: append the hoisted fixed-parameter
        sub-expressions (evaluated here, once per call, in float64)."""
        self._native_exprs()
        pr = np.asarray(pr, dtype=np.float64)
        if not self._hoisted and not self._packed and not self._mv_index:
            return pr
        lead = pr.shape[:-1]
        flat = pr.reshape(-1, self.n_remainder)
        pieces = [flat]
        if self._hoisted:
            if self._hoist_fn is None:
                self._hoist_fn = sym.lambdify([list(self._sym_fixed_paramsvec)], self._hoisted,
                                              modules=[_HOST_HELPERS, "numpy"], cse=True)
            with np.errstate(all="ignore"):
                extra = np.array([np.asarray(self._hoist_fn(list(row)), dtype=np.float64) for row in flat])
            pieces.append(extra.reshape(len(flat), -1))
        if self._packed:
            pieces.append(flat[:, np.asarray(self._packed, dtype=np.int64)])
        if self._mv_index:          # j-major, sign-folded copies of the matrix-vector blocks (may read hoisted values)
            src = np.concatenate(pieces[:2], axis=1) if self._hoisted else flat
            idx = np.asarray(self._mv_index, dtype=np.int64)
            sign = np.asarray(self._mv_sign, dtype=np.float64)
            pieces.append(np.where(idx >= 0, src[:, np.maximum(idx, 0)], 0.0) * sign)
        return np.concatenate(pieces, axis=1).reshape(lead + (-1,))

    def native_source(self) -> str:
        """Generated header with the five callbacks (HIP ``__device__`` and host C)."""
        if self._native_source is None:
            desc = "states=%s params=%s deriv=%s" % (
                [".".join(p) for p in self.state_subset.paths],
                [".".join(p) for p in self.params_subset.paths],
                [".".join(p) for p in self.params_subset.subset_paths])
            (dydt, jac, dlamdadt, quad, dydp_t), slots = self._native_exprs()
            if self._hoisted or self._packed or self._matvec:
                desc += " hoisted=%d packed=%d matvec=%s matfill=%s" % (
                    len(self._hoisted), len(self._packed),
                    {k: (v["n_out"], v["n_in"]) for k, v in self._matvec.items()},
                    {k: (v["axis"], len(v["exceptions"])) for k, v in self._matfill.items()})
            self._native_source = codegen.generate_problem_source(
                n_states=self.n_states, n_sub=self.n_params, n_rem=self.n_remainder_native,
                symbol_map=slots, dydt=dydt, jac=jac, dlamdadt=dlamdadt, quad=quad, dydp_t=dydp_t,
                description=desc, matvec=self._matvec, matfill=self._matfill,
                leaf_axes=None if os.environ.get("SA_NO_LANE_FAMILIES") else self._leaf_axes(),
            )
        return self._native_source

    # ------------------------------------------------------------------
    # host callables with the reference's signatures (reference :251-433)
    def _host(self, key: str, expr: np.ndarray, with_lamda: bool):
        if key not in self._host_funcs:
            args = [self._sym_time, list(self._sym_statevec)]
            if with_lamda:
                args.append(list(self._sym_lamda))
            args += [list(self._sym_deriv_paramsvec), list(self._sym_fixed_paramsvec)]
            flat = [sym.sympify(e) for e in np.asarray(expr, dtype=object).ravel()]
            fn = sym.lambdify(args, flat, modules=[_HOST_HELPERS, "numpy"], cse=True)
            self._host_funcs[key] = (fn, np.asarray(expr, dtype=object).shape)
        return self._host_funcs[key]

    def _eval_host(self, key, expr, with_lamda, out, t, y, lamda, user_data):
        fn, shape = self._host(key, expr, with_lamda)
        yv = np.asarray(y).reshape(1).view(np.float64) if np.asarray(y).dtype.fields else np.asarray(y, float)
        ps, pr = self.flat_params(user_data) if user_data is not None else (np.zeros(0), np.zeros(0))
        with np.errstate(all="ignore"):
            if with_lamda:
                vals = fn(float(t), list(yv), list(np.asarray(lamda, float)), list(ps), list(pr))
            else:
                vals = fn(float(t), list(yv), list(ps), list(pr))
        out[...] = np.asarray(vals, dtype=float).reshape(shape)
        return 0 if np.isfinite(out).all() else 1

    def make_rhs(self, *, debug=False):
        def rhs(out, t, y, user_data):
            code = self._eval_host("rhs", self._sym_dydt, False, out, t, y, None, user_data)
            if code and user_data is not None:
                user_data.error_rhs[:] = out
            return code
        return rhs

    def make_adjoint_rhs(self, *, debug=False):
        def adjoint(out, t, y, lamda, user_data):
            return self._eval_host("adj", self._sym_dlamdadt, True, out, t, y, lamda, user_data)
        return adjoint

    def make_adjoint_quad_rhs(self, *, debug=False):
        def quad_rhs(out, t, y, lamda, user_data):
            return self._eval_host("quad", self._sym_quad_rhs, True, out, t, y, lamda, user_data)
        return quad_rhs

    def make_jac_dense(self, *, debug=False):
        def jac_dense(out, t, y, fy, user_data):
            return self._eval_host("jac", self._sym_dydt_jac, False, out, t, y, None, user_data)
        return jac_dense

    def make_adjoint_jac_dense(self, *, debug=False):
        def jac_dense(out, t, y, yB, fyB, user_data):
            return self._eval_host("adjjac", -self._sym_dydt_jac.T, False, out, t, y, None, user_data)
        return jac_dense


def _logaddexp(a, b):
    return np.logaddexp(a, b)


def _expit(x):
    return 1.0 / (1.0 + np.exp(-x))


def _cardinal_bspline(degree, t):
    """Cardinal B-spline of any degree by the Cox-de Boor recursion on the knots 0..degree+1 (host evaluation)."""
    t = np.asarray(t, dtype=float)
    d = int(degree)
    basis = [((t >= i) & (t < i + 1)).astype(float) for i in range(d + 1)]
    for k in range(1, d + 1):
        basis = [((t - i) * basis[i] + (i + k + 1 - t) * basis[i + 1]) / k for i in range(d + 1 - k)]
    return basis[0]


_HOST_HELPERS = {
    "logaddexp": _logaddexp,
    "expit": _expit,
    "dexpit": lambda x: _expit(x) * _expit(-x),
    "CardinalBSpline": _cardinal_bspline,
}
#: for ``sympy.lambdify(..., modules=[HOST_FUNCTIONS, "numpy"])`` of expressions that use symode/lambdify.py's functions
HOST_FUNCTIONS = _HOST_HELPERS
