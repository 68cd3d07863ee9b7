"""``Solver`` / ``AdjointSolver`` with the reference's signatures, backed by the HIP engine.

Mirrors /root/reference/sunode/solver.py: ``Solver`` (``:213-527``), ``AdjointSolver``
(``:530-784``), ``SolverError`` (``:21``).  Where the reference holds one CVODES memory
block and integrates one parameter set per call, these classes hold one ``sa_solver``
handle (C ABI, include/sunode_amd.h) and integrate a whole batch per call; the scalar
methods of the reference API are the B = 1 case of the same kernels.

Scope: BDF with dense Newton/LU and analytic Jacobians in both
directions, polynomial interpolation of the stored forward trajectory.  Options of the
reference outside that path raise ``NotImplementedError`` instead of silently doing
something else.  There is no CPU fallback.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import numpy as np

from sunode_amd import _native
from sunode_amd.dtypesubset import _as_dict

#: CVODES return codes -> names (reference builds this table from the cffi lib, basic.py:49-55)
ERRORS: Dict[int, str] = {
    0: "CV_SUCCESS", 1: "CV_TSTOP_RETURN", 2: "CV_ROOT_RETURN", 99: "CV_WARNING",
    -1: "CV_TOO_MUCH_WORK", -2: "CV_TOO_MUCH_ACC", -3: "CV_ERR_FAILURE", -4: "CV_CONV_FAILURE",
    -5: "CV_LINIT_FAIL", -6: "CV_LSETUP_FAIL", -7: "CV_LSOLVE_FAIL", -8: "CV_RHSFUNC_FAIL",
    -9: "CV_FIRST_RHSFUNC_ERR", -10: "CV_REPTD_RHSFUNC_ERR", -11: "CV_UNREC_RHSFUNC_ERR",
    -12: "CV_RTFUNC_FAIL", -13: "CV_NLS_INIT_FAIL", -14: "CV_NLS_SETUP_FAIL", -15: "CV_CONSTR_FAIL",
    -20: "CV_MEM_FAIL", -21: "CV_MEM_NULL", -22: "CV_ILL_INPUT", -23: "CV_NO_MALLOC", -24: "CV_BAD_K",
    -25: "CV_BAD_T", -26: "CV_BAD_DKY", -27: "CV_TOO_CLOSE", -28: "CV_VECTOROP_ERR",
    -30: "CV_NO_QUAD", -31: "CV_QRHSFUNC_FAIL", -32: "CV_FIRST_QRHSFUNC_ERR",
    -33: "CV_REPTD_QRHSFUNC_ERR", -34: "CV_UNREC_QRHSFUNC_ERR",
    -40: "CV_NO_SENS", -41: "CV_SRHSFUNC_FAIL", -42: "CV_FIRST_SRHSFUNC_ERR", -43: "CV_REPTD_SRHSFUNC_ERR",
    -44: "CV_UNREC_SRHSFUNC_ERR", -45: "CV_BAD_IS",
    -101: "CV_NO_ADJ", -102: "CV_NO_FWD", -103: "CV_NO_BCK", -104: "CV_BAD_TB0",
    -105: "CV_REIFWD_FAIL", -106: "CV_FWD_FAIL", -107: "CV_GETY_BADT",
    # not a CVODES code: the stored forward trajectory of the instance exceeds max_steps points (or 64 such
    # instances exceed the arena budget) -- raise AdjointSolver(max_steps=..., arena_gib=...)
    -9001: "SA_STATUS_ARENA_FULL",
}


class SolverError(RuntimeError):
    pass


def initial_sensitivities(problem) -> np.ndarray:
    """sens0 [n_params, n_states] of the forward-sensitivity Op (reference wrappers/as_pytensor.py:201-230): zero, except that a
    differentiated parameter living under the top-level key ``__initial_values`` IS an initial value --
    its sensitivity starts as the unit vector of the state entry with the same path."""
    offsets, pos = {}, 0
    for path in problem.state_subset.paths:
        offsets[path] = pos
        pos += int(np.prod(problem.state_subset.flat_shapes[path], dtype=int))
    sens0 = np.zeros((problem.n_params, problem.n_states))
    row = 0
    for path in problem.params_subset.subset_paths:
        n_items = int(np.prod(problem.params_subset.flat_shapes[path], dtype=int))
        if path and path[0] == "__initial_values":
            state_path = tuple(path[1:])
            if state_path not in offsets:
                raise ValueError("no state with path %s for initial-value parameter" % (state_path,))
            for k in range(n_items):
                sens0[row + k, offsets[state_path] + k] = 1.0
        row += n_items
    return sens0


def _flat_state(problem, y0) -> np.ndarray:
    y0 = np.asarray(y0)
    if y0.dtype == problem.state_dtype:
        y0 = y0[None].view(np.float64)
    y0 = np.ascontiguousarray(y0, dtype=np.float64).reshape(-1)
    if y0.shape != (problem.n_states,):
        raise ValueError(f"y0 should have shape {(problem.n_states,)} but has shape {y0.shape}.")
    return y0


def _rows(a: np.ndarray, lo: int, hi: int, per_instance) -> np.ndarray:
    """Rows [lo, hi) of a per-instance array; arrays shared by the batch (``per_instance`` false: no differentiated
    parameters / a shared remainder vector) are passed whole."""
    return a[lo:hi] if per_instance else a


class _OutputPool:
    """Output arrays of the batch methods, recycled -- OPT-IN (``reuse_outputs=True``).

    A FRESH output array of BASELINE config 2's ``y_out`` (52 MB) costs more than its PCIe transfer: every page is
    faulted in by the copy that first touches it -- tools/ubench_hostcopy.py on the MI355X box: device -> host into a
    fresh array 4.4 ms, into one that was touched before 1.0 ms (= the pinned rate; the runtime pins on the fly).
    The reference's convention avoids that by construction: the CALLER allocates the outputs once and the solver
    writes in place (/root/reference/sunode/solver.py:682,723-724) -- the batch methods take ``out=`` for exactly that,
    and it is the recommended way.  With ``reuse_outputs=True`` the solver does it for the caller: ONE array per
    (method slot, shape, dtype), handed out again by the next call of the same method.  The contract is explicit
    (no reference counting: a caller may hold an array through a raw pointer the interpreter cannot see): a result
    of a batch call is valid until the next call of the same method on the same solver; copy what must live longer.
    Without the option every call returns fresh arrays."""

    MAX_BYTES = 1 << 31

    def __init__(self):
        self._arrays = {}
        self._bytes = 0

    def get(self, slot, shape, dtype=np.float64):
        key = (slot, tuple(int(k) for k in shape), np.dtype(dtype).str)
        arr = self._arrays.get(key)
        if arr is None:
            arr = np.empty(key[1], dtype=dtype)
            if self._bytes + arr.nbytes <= self.MAX_BYTES:
                self._arrays[key] = arr
                self._bytes += arr.nbytes
        return arr


def _is_device_tensor(x) -> bool:
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


class _EngineMixin:
    """Shared plumbing: native handle (lazy), parameter record, batch argument checks."""

    _problem: Any
    _user_data: np.ndarray

    def _native_kwargs(self) -> Dict[str, Any]:
        raise NotImplementedError

    def _engine_kwargs(self) -> Dict[str, Any]:
        """Keyword arguments of one NativeSolver handle (without device / arena share)."""
        return self._native_kwargs()

    # -- devices (the batch shards by instance, no exchange step) ------------
    # ``devices=[0, 1, ..., 7]``: ONE Python process drives several GPUs -- the reference's call pattern is one solver
    # object inside one PyMC process (/root/reference/sunode/wrappers/as_pytensor.py:279-344).  One ``sa_solver``
    # handle (own stream, own trajectory arena) per entry; every batch call splits the instances into contiguous
    # balanced shards (parallel.shard_bounds), issues the shards from one thread per handle (ctypes releases the
    # GIL for the duration of the native call) and the handles write into disjoint row ranges of the caller's
    # arrays.  An ordinal may appear more than once (two handles on one device: the arena budget of the device is
    # split between them).  Results are those of the one-handle call bit for bit (instances are independent).
    def _init_devices(self, device, devices, interleaved=False):
        self._interleaved = bool(interleaved)
        if devices is None:
            devices = [int(device)]
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("devices must name at least one HIP device ordinal")
        if any(d < 0 for d in devices):
            raise ValueError("negative device ordinal")
        self._devices = devices
        self._device = devices[0]
        self._natives = None
        self._native_sets = {}
        self._mapping = None
        self._pool = None
        self._outputs = _OutputPool() if getattr(self, "_reuse_outputs", False) else None

    def _out(self, slot, shape, dtype=np.float64, given=None, small=False):
        """(work, final) arrays of one output of a batch call.  ``final`` is what the caller receives: the array it
        passed through ``out=`` (checked: exact shape, dtype, C-contiguous, writeable -- the library writes through
        the raw pointer), else a fresh one (or the solver's recycled one, ``reuse_outputs=True``).  ``work`` is what
        the library writes: ``final`` itself, or -- with ``interleaved=True`` over several handles -- a handle-major
        temporary that ``_scatter`` copies into ``final``.  ``small`` outputs (status, counters) start zeroed: the
        library returns early without touching them for an empty time grid."""
        shape = tuple(int(k) for k in shape)
        dtype = np.dtype(dtype)

        def fresh(tag):
            pool = getattr(self, "_outputs", None)
            arr = pool.get((slot, tag), shape, dtype) if pool is not None else np.empty(shape, dtype=dtype)
            if small:
                arr.fill(0)
            return arr
        if given is not None:
            if not isinstance(given, np.ndarray) or given.shape != shape or given.dtype != dtype \
                    or not given.flags.c_contiguous or not given.flags.writeable:
                raise ValueError("out[%r] must be a writeable C-contiguous %s array of shape %s" % (slot, dtype, shape))
            if small:
                given.fill(0)
            final = given
        else:
            final = fresh("final")
        work = fresh("work") if self._reorders() else final
        return work, final

    @staticmethod
    def _zero_width(given, shape):
        if given is not None and tuple(given.shape) != tuple(shape):
            raise ValueError("out= array of shape %s where %s is expected" % (given.shape, shape))
        return given if given is not None else np.zeros(shape)

    @staticmethod
    def _out_arg(out, k: int, names):
        """``out=`` of the batch methods: None, a dict keyed by the output's name, or a sequence in return order
        (entries may be None: allocated by the solver)."""
        if out is None:
            return None
        if isinstance(out, dict):
            unknown = set(out) - set(names)
            if unknown:
                raise ValueError("out= has unknown entries %s (expected some of %s)" % (sorted(unknown), list(names)))
            return out.get(names[k])
        if len(out) > len(names):
            raise ValueError("out= has %d entries, the call returns %d arrays" % (len(out), len(names)))
        return out[k] if k < len(out) else None

    def _arena_share(self, device: int) -> int:
        """arena_bytes of ONE handle on ``device``: the device's budget divided by the handles that share it.
        With one handle per device the library's own default applies (0)."""
        sharing = self._devices.count(device)
        budget = getattr(self, "_arena_bytes", 0)
        if sharing == 1:
            return budget
        if not budget:          # the library default, evaluated once for the device instead of once per handle
            free_b, _total = _native.device_memory(device)
            budget = min(96 << 30, int(0.6 * free_b))
        return max(budget // sharing, 1)

    def _engines(self):
        """The handles of the ACTIVE mapping (one per entry of ``devices``), created on first use.  ``self._mapping`` is
        None for the engine's own choice of kernel family, or a ``group`` value of ``_native.kernel_variant`` (the
        small-batch mapping of an AdjointSolver, ``_select_mapping``): each mapping has its own code object and handles."""
        sets = self.__dict__.setdefault("_native_sets", {})
        key = getattr(self, "_mapping", None)
        if key not in sets:
            kw = self._engine_kwargs()
            if key is not None:
                kw["group"] = key
            natives = []
            for d in self._devices:
                k = dict(kw, device=d)
                if "arena_bytes" in k:
                    k["arena_bytes"] = self._arena_share(d)
                natives.append(_native.NativeSolver(self._source, n_states=self._problem.n_states, **k))
            sets[key] = natives
        self._natives = sets[key]
        self._native = self._natives[0]
        return self._natives

    def _engine(self) -> _native.NativeSolver:
        return self._engines()[0]

    def _select_mapping(self, B: int) -> None:
        """Activate the handles this batch runs on: more lanes per instance while every handle's share of the batch is
        small enough for the chip to hold them all at once (``_native.small_batch_group``), the engine's own choice
        otherwise.  Bit-identical results either way."""
        per_handle = max(1, -(-int(B) // len(self._devices)))
        self._mapping = _native.small_batch_group(self._source, getattr(self, "_hermite", False), per_handle) \
            if getattr(self, "_batch_mapping", "fixed") == "auto" else None

    def _shards(self, B: int):
        """[(handle, lo, hi)] -- contiguous balanced instance ranges, empty ones dropped.  With ``interleaved=True``
        the ranges refer to the batch in HANDLE-MAJOR order (see _gather / _scatter)."""
        from sunode_amd.parallel import shard_bounds
        engines = self._engines()
        G = len(engines)
        out = []
        pos = 0
        for r, eng in enumerate(engines):
            if self._interleaved and G > 1:
                k = len(range(r, B, G))
                lo, hi = pos, pos + k
                pos = hi
            else:
                lo, hi = shard_bounds(B, r, G)
            if hi > lo:
                out.append((eng, lo, hi))
        return out

    # ``interleaved=True`` ("if per-instance cost varies strongly (Robertson), interleaved assignment
    # i mod G balances better than contiguous; keep output indexing stable"): handle r integrates the instances
    # r, r + G, r + 2G, ...  The C entry points take contiguous arrays, so per-instance inputs are gathered into
    # handle-major order once per call (strided copies) and the results scattered back to the caller's order.
    def _reorders(self) -> bool:
        return self._interleaved and len(self._engines()) > 1

    def _gather(self, a: np.ndarray, per_instance=True) -> np.ndarray:
        if not per_instance or not self._reorders():
            return a
        G = len(self._engines())
        return np.concatenate([a[r::G] for r in range(G)], axis=0)

    def _scatter(self, pair) -> np.ndarray:
        """(work, final) of ``_out`` after the call: ``final`` in the caller's instance order."""
        handle_major, out = pair
        if out is handle_major:
            return out
        G = len(self._engines())
        B = handle_major.shape[0]
        pos = 0
        for r in range(G):
            k = len(range(r, B, G))
            out[r::G] = handle_major[pos:pos + k]
            pos += k
        return out

    def _run_shards(self, shards, call):
        """call(handle, lo, hi) for every shard: inline for one, one thread per handle otherwise; the first
        exception (in shard order) propagates after all shards have finished."""
        if len(shards) == 1:
            call(*shards[0])
            return
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=len(self._devices), thread_name_prefix="sunode_amd")
        futures = [self._pool.submit(call, *sh) for sh in shards]
        errors = []
        for f in futures:
            try:
                f.result()
            except Exception as exc:        # noqa: BLE001 -- re-raised below
                errors.append(exc)
        if errors:
            raise errors[0]

    def last_kernel_ms(self):
        """(forward, backward) kernel time of the last batch call: the slowest handle's (they run concurrently)."""
        ms = [e.last_kernel_ms() for e in self._engines()]
        return max(m[0] for m in ms), max(m[1] for m in ms)

    def _set_retries(self, **kw):
        for eng in self._engines():
            if any(eng._opt_kw[k] != v for k, v in kw.items()):
                eng.set_options(**kw)

    # -- reference parameter API (solver.py:435-465 / 650-680) ----------------------------
    @property
    def params_dtype(self):
        return self._problem.params_dtype

    @property
    def derivative_params_dtype(self):
        return self._problem.params_subset.subset_dtype

    @property
    def remainder_params_dtype(self):
        return self._problem.params_subset.remainder.subset_dtype

    def set_params(self, params):
        self._problem.update_params(self._user_data, params)

    def get_params(self):
        return self._problem.extract_params(self._user_data)

    def set_params_dict(self, params):
        data = self.get_params()
        self._problem.params_subset.from_dict(params, data)
        self.set_params(data)

    def get_params_dict(self):
        return _as_dict(self.get_params())

    def set_derivative_params(self, params):
        self._problem.update_subset_params(self._user_data, params)

    def set_remaining_params(self, params):
        self._problem.update_remaining_params(self._user_data, params)

    def as_xarray(self, tvals, out, sens_out=None, unstack_state=True, unstack_params=True):
        return self._problem.solution_to_xarray(
            tvals, out, self._user_data, sensitivity=sens_out,
            unstack_state=unstack_state, unstack_params=unstack_params)

    # -- batch helpers ---------------------------------------------------------------------
    def _batch_inputs(self, y0, params_sub, params_rem):
        n, p, r = self._problem.n_states, self._problem.n_params, self._problem.n_remainder
        y0 = np.ascontiguousarray(y0, dtype=np.float64)
        if y0.ndim != 2 or y0.shape[1] != n:
            raise ValueError(f"y0 must have shape (B, {n})")
        B = y0.shape[0]
        ps = np.ascontiguousarray(params_sub, dtype=np.float64).reshape(B, p) if p else np.zeros((B, 0))
        pr = np.ascontiguousarray(params_rem, dtype=np.float64) if r else np.zeros(0)
        if r and pr.shape == (r,):
            stride = 0
        elif r and pr.shape == (B, r):
            stride = self._problem.n_remainder_native
        elif r:
            raise ValueError(f"params_rem must have shape ({r},) or (B, {r})")
        else:
            stride = 0
        if r:       # hoisted fixed-parameter sub-expressions ride at the end of the remainder vector
            pr = np.ascontiguousarray(self._problem.extend_remainder(pr))
        if ps.size == 0:
            ps = np.zeros(1)
        if pr.size == 0:
            pr = np.zeros(1)
        return B, y0, ps, pr, stride

    @staticmethod
    def stats_as_dict(stats: np.ndarray) -> Dict[str, np.ndarray]:
        return {name: stats[..., i] for i, name in enumerate(_native.STAT_NAMES[:15])}


class Solver(_EngineMixin):
    """Forward solves (reference ``Solver``, solver.py:213-527) -- BDF/dense only."""

    def __init__(self, problem, *, abstol: float = 1e-10, reltol: float = 1e-10, sens_mode: Optional[str] = None,
                 scaling_factors: Optional[np.ndarray] = None, constraints: Optional[np.ndarray] = None,
                 solver="BDF", linear_solver="dense", linear_solver_kwargs=None, mxsteps: int = 500,
                 device: int = 0, devices=None, interleaved: bool = False, reuse_outputs: bool = False,
                 batch_mapping: str = "auto"):
        if batch_mapping not in ("auto", "fixed"):
            raise ValueError('batch_mapping must be "auto" or "fixed"')
        if sens_mode in (None, False):
            sens_mode = None
        elif sens_mode == "staggered1":
            raise ValueError("staggered1 not implemented.")            # as the reference, solver.py:365-366
        elif sens_mode not in ("simultaneous", "staggered"):
            raise ValueError('sens_mode must be one of "simultaneous" and "staggered".')
        if scaling_factors is not None:
            scaling_factors = np.ascontiguousarray(scaling_factors, dtype=np.float64)
            if scaling_factors.shape != (problem.n_params,):
                raise ValueError("Invalid shape of scaling_factors.")
        if solver != "BDF":
            if solver == "ADAMS":
                raise NotImplementedError("only the BDF method is implemented on the device")
            raise ValueError(f"Unknown solver {solver}.")
        if linear_solver not in ("dense", "dense_finitediff", "spgmr", "spgmr_finitediff", "band"):
            raise ValueError(f"Unknown linear solver: {linear_solver}")
        if linear_solver != "dense":
            # The reference offers these as alternative ways to solve the same Newton systems (solver.py:326-358).
            # The device integrator has one: dense LU of I - gamma*J with the analytic Jacobian.  The option is
            # accepted so that reference code runs unchanged; results agree to the integration tolerance.
            import warnings
            warnings.warn(f"linear_solver={linear_solver!r}: the MI355X integrator always uses dense LU with the "
                          "analytic Jacobian", RuntimeWarning, stacklevel=2)
        if constraints is not None:
            constraints = np.broadcast_to(np.asarray(constraints, dtype=np.float64), (problem.n_states,)).copy()
            if not np.isin(constraints, (0.0, 1.0, -1.0, 2.0, -2.0)).all():
                raise ValueError("constraints entries must be 0, +-1 or +-2 (CVodeSetConstraints)")
            if sens_mode == "simultaneous":
                raise ValueError("constraints can not be enforced with the simultaneous sensitivity corrector")
            if sens_mode is not None:
                raise NotImplementedError("constraints together with forward sensitivities")
        self._problem = problem
        self._user_data = problem.make_user_data()
        self._constraints = constraints
        self._abstol, self._reltol = abstol, reltol
        self._linear_solver_kind = linear_solver
        self._linear_solver_kwargs = linear_solver_kwargs or {}
        self._sens_mode = sens_mode
        self._compute_sens = sens_mode is not None
        self._scaling_factors = scaling_factors
        self._mxsteps = mxsteps
        self._reuse_outputs = bool(reuse_outputs)
        # (the sensitivity builds keep one mapping: their lane-group form exists for 4 / 8 lanes only)
        self._batch_mapping = batch_mapping if sens_mode is None else "fixed"
        self._init_devices(device, devices, interleaved)
        self._set_tolerances(abstol, reltol)
        self._state_names = ["_problem", "_user_data", "_constraints", "_abstol", "_reltol", "_batch_mapping",
                             "_linear_solver_kind", "_linear_solver_kwargs", "_sens_mode", "_scaling_factors",
                             "_mxsteps", "_device", "_devices", "_interleaved", "_reuse_outputs", "_state_names"]
        self._init_native()

    def _init_native(self):
        self._source = self._problem.native_source()
        # compile at construction like the reference JITs; sensitivity solves use their own build
        _native.build_code_object(self._source, sens=self._compute_sens, constraints=self._constraints is not None)
        self._native = None
        self._natives = None
        self._native_sets = {}
        self._mapping = None
        self._pool = None

    def _engine_kwargs(self):
        return dict(self._native_kwargs(), sens=self._compute_sens, constraints=self._constraints,
                    guard_kinds=("sens",) if self._compute_sens else ("plain",))

    def __getstate__(self):
        return {name: self.__dict__[name] for name in self._state_names}

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._reuse_outputs = state.get("_reuse_outputs", False)
        self._batch_mapping = state.get("_batch_mapping", "auto" if self._sens_mode is None else "fixed")
        self._outputs = _OutputPool() if self._reuse_outputs else None
        self._compute_sens = self._sens_mode is not None
        self._set_tolerances(self._abstol, self._reltol)
        self._init_native()

    def _set_tolerances(self, atol=None, rtol=None):
        atol, rtol = np.array(atol, dtype=float), np.array(rtol, dtype=float)
        if rtol.ndim != 0:
            raise ValueError("vector reltol is not supported (the reference's CVodeVVtolerances path is "
                             "undeclared in its cdef headers)")
        if atol.ndim == 1 and atol.shape != (self._problem.n_states,):
            raise ValueError("Invalid tolerance.")
        if atol.ndim > 1:
            raise ValueError("Invalid tolerance.")
        self._atol, self._rtol = atol, rtol

    def _native_kwargs(self):
        return dict(rtol=float(self._rtol), atol=self._atol, mxstep=self._mxsteps, traj_capacity=2)

    def make_output_buffers(self, tvals):
        """y_out, or (y_out, sens_out) when sensitivities are computed (reference solver.py:419-426)."""
        n, p = self._problem.n_states, self._problem.n_params
        y_out = np.zeros((len(tvals), n))
        if self._compute_sens:
            return y_out, np.zeros((len(tvals), p, n))
        return y_out

    def solve(self, t0, tvals, y0, y_out, *, sens0=None, sens_out=None, max_retries=5):
        if self._compute_sens and (sens0 is None or sens_out is None):
            raise ValueError('"sens_out" and "sens0" are required when computin sensitivities.')
        y0 = _flat_state(self._problem, y0)
        ps, pr = self._problem.flat_params(self._user_data)
        tvals = np.ascontiguousarray(tvals, dtype=np.float64)
        if self._compute_sens:
            yo, so, status, _ = self.solve_sens_batch(t0, tvals, y0[None], ps[None], pr,
                                                      np.asarray(sens0, dtype=np.float64)[None],
                                                      max_retries=max_retries)
        else:
            yo, status, _ = self.solve_batch(t0, tvals, y0[None], ps[None], pr, max_retries=max_retries)
        if status[0] != 0:
            code = int(status[0])
            if code == -1:
                raise SolverError("Too many solver retries.")
            raise SolverError(f"Solving ode failed: {ERRORS.get(code, 'unknown')} ({code})")
        y_out[...] = yo[0]
        if self._compute_sens:
            sens_out[...] = so[0]

    def solve_sens_batch(self, t0, tvals, y0, params_sub, params_rem, sens0, *, max_retries=5, out=None
                         ) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        """Forward solve + forward sensitivities for B draws (``sens_mode`` must be set): returns
        (y_out [B,n_t,n], sens_out [B,n_t,p,n], status [B], stats [B,16]); ``sens0`` is [B,p,n] or [p,n].
        ``out``: caller-allocated outputs (see ``solve_batch``), names ``y_out, sens_out, status, stats``."""
        if not self._compute_sens:
            raise ValueError("construct the Solver with sens_mode='simultaneous' or 'staggered'")
        self._set_retries(max_retries_fwd=max_retries)
        B, y0, ps, pr, stride = self._batch_inputs(y0, params_sub, params_rem)
        n, p = self._problem.n_states, self._problem.n_params
        sens0 = np.ascontiguousarray(np.broadcast_to(np.asarray(sens0, dtype=np.float64), (B, p, n)))
        tvals = np.ascontiguousarray(tvals, dtype=np.float64)
        y0, ps, pr, sens0 = self._gather(y0), self._gather(ps, p), self._gather(pr, stride), self._gather(sens0)
        names = ("y_out", "sens_out", "status", "stats")
        y_pair = self._out("sens.y_out", (B, len(tvals), n), given=self._out_arg(out, 0, names))
        s_pair = self._out("sens.sens_out", (B, len(tvals), p, n), given=self._out_arg(out, 1, names))
        st_pair = self._out("sens.status", (B,), np.int32, self._out_arg(out, 2, names), small=True)
        sa_pair = self._out("sens.stats", (B, _native.N_STATS), np.int64, self._out_arg(out, 3, names), small=True)
        y_out, sens_out, status, stats = y_pair[0], s_pair[0], st_pair[0], sa_pair[0]
        ism = 0 if self._sens_mode == "simultaneous" else 1

        def call(eng, lo, hi):
            eng.solve_sens(_native.SA_MEM_HOST, ism, self._scaling_factors, hi - lo, y0[lo:hi],
                           _rows(ps, lo, hi, p), _rows(pr, lo, hi, stride), stride,
                           sens0[lo:hi] if sens0.size else np.zeros(1), t0, tvals, len(tvals), y_out[lo:hi],
                           sens_out[lo:hi] if sens_out.size else np.zeros(1), status[lo:hi], stats[lo:hi])
        self._run_shards(self._shards(B), call)
        return self._scatter(y_pair), self._scatter(s_pair), self._scatter(st_pair), self._scatter(sa_pair)

    def solve_batch(self, t0, tvals, y0, params_sub, params_rem, *, max_retries=5, out=None
                    ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """``solve`` for B parameter draws at once: returns (y_out [B,n_t,n], status [B], stats [B,16]).

        ``out``: caller-allocated output arrays, written in place and returned -- the reference's convention
        (``solve(t0, tvals, y0, y_out)``, /root/reference/sunode/solver.py:467) carried over to the batch: a dict
        ``{"y_out": ..., "status": ..., "stats": ...}`` or a sequence in return order, entries may be missing / None.
        Arrays must be C-contiguous float64 (status int32, stats int64) of the exact shape."""
        self._select_mapping(np.shape(y0)[0] if np.ndim(y0) == 2 else 0)
        self._set_retries(max_retries_fwd=max_retries)
        B, y0, ps, pr, stride = self._batch_inputs(y0, params_sub, params_rem)
        tvals = np.ascontiguousarray(tvals, dtype=np.float64)
        p = self._problem.n_params
        y0, ps, pr = self._gather(y0), self._gather(ps, p), self._gather(pr, stride)
        names = ("y_out", "status", "stats")
        y_pair = self._out("solve.y_out", (B, len(tvals), self._problem.n_states), given=self._out_arg(out, 0, names))
        st_pair = self._out("solve.status", (B,), np.int32, self._out_arg(out, 1, names), small=True)
        sa_pair = self._out("solve.stats", (B, _native.N_STATS), np.int64, self._out_arg(out, 2, names), small=True)
        y_out, status, stats = y_pair[0], st_pair[0], sa_pair[0]

        def call(eng, lo, hi):
            eng.solve(_native.SA_MEM_HOST, hi - lo, y0[lo:hi], _rows(ps, lo, hi, p), _rows(pr, lo, hi, stride),
                      stride, t0, tvals, len(tvals), y_out[lo:hi], status[lo:hi], stats[lo:hi])
        self._run_shards(self._shards(B), call)
        return self._scatter(y_pair), self._scatter(st_pair), self._scatter(sa_pair)


class AdjointSolver(_EngineMixin):
    """Forward + adjoint solves (reference ``AdjointSolver``, solver.py:530-784).

    Extra keyword arguments (not in the reference): ``backward_abstol/backward_reltol`` and
    ``quad_abstol/quad_reltol`` expose what the reference hard-codes to 1e-10
    (solver.py:599,614 -- there one has to poke ``lib.CVodeSStolerancesB`` by hand,
    README.md:245-247); ``max_steps`` is the most stored forward steps one instance may take
    (default ``checkpoint_n + 1``, i.e. as unbounded as the reference) and ``arena_gib`` the HBM
    budget of the stored trajectories (default 96 GiB, at most 60 % of the free memory): batches
    that fit stay resident between ``solve_forward`` and ``solve_backward``, larger ones are
    re-integrated tile by tile inside ``solve_backward`` -- CVODES' check-point scheme, same
    results (csrc/sunode_amd.cpp, "trajectory arena").  ``compact_trajectory=True`` stores what CVODES itself
    keeps per step, {order, t, y[n]}, instead of the ready-made interpolation table (8 + 6n doubles) and lets the
    backward kernel rebuild the table when its index moves: 5x less arena and HBM traffic (one-lane-per-instance
    kernel, polynomial interpolation; ignored elsewhere); results identical.  Default (None): on from three states
    (Robertson B = 262 144: 73 -> 14 GB, 2.37 -> 2.79 M solves/s; Lotka-Volterra, n = 2, is 4 % faster without).
    """

    def __init__(self, problem, *, abstol=1e-10, reltol=1e-10, checkpoint_n=500_000, interpolation="polynomial",
                 constraints=None, solver="BDF", adjoint_solver="BDF", backward_abstol=1e-10,
                 backward_reltol=1e-10, quad_abstol=1e-10, quad_reltol=1e-10, mxsteps: int = 500,
                 max_steps: Optional[int] = None, arena_gib: Optional[float] = None, device: int = 0,
                 compact_trajectory: Optional[bool] = None, devices=None, interleaved: bool = False,
                 reuse_outputs: bool = False, batch_mapping: str = "auto"):
        if batch_mapping not in ("auto", "fixed"):
            raise ValueError('batch_mapping must be "auto" or "fixed"')
        if solver not in ("BDF", "ADAMS"):
            raise ValueError(f"Unknown solver {solver}.")
        if adjoint_solver not in ("BDF", "ADAMS"):
            raise ValueError(f"Unknown solver {adjoint_solver}.")
        if solver != "BDF" or adjoint_solver != "BDF":
            raise NotImplementedError("only the BDF method is implemented on the device")
        if interpolation not in ("polynomial", "hermite"):
            raise ValueError(f"Unknown interpolation {interpolation}.")
        self._hermite = interpolation == "hermite"
        if constraints is not None:           # forward problem only, as in the reference (solver.py:569-572)
            constraints = np.broadcast_to(np.asarray(constraints, dtype=np.float64), (problem.n_states,)).copy()
            if not np.isin(constraints, (0.0, 1.0, -1.0, 2.0, -2.0)).all():
                raise ValueError("constraints entries must be 0, +-1 or +-2 (CVodeSetConstraints)")
        self._problem = problem
        self._user_data = problem.make_user_data()
        self._constraints = constraints
        self._set_tolerances(abstol, reltol)
        self._tolB = (float(backward_reltol), float(backward_abstol), float(quad_reltol), float(quad_abstol))
        self._mxsteps = mxsteps
        self._max_steps = int(min(max_steps if max_steps is not None else checkpoint_n + 1, checkpoint_n + 1,
                                  2**31 - 1))
        self._arena_bytes = int(arena_gib * 2**30) if arena_gib else 0      # per DEVICE (handles sharing one split it)
        self._reuse_outputs = bool(reuse_outputs)
        self._init_devices(device, devices, interleaved)
        self._source = problem.native_source()
        # (bdf_kernels.hip and bdf_wave.hip carry the compact-record option)
        if compact_trajectory is None:        # measured (profiles/r03_compact_trajectory.txt): pays from three states on
            compact_trajectory = _native.default_compact_trajectory(self._source, self._hermite)
        self._compact = bool(compact_trajectory) and not self._hermite and \
            _native.kernel_variant(self._source, hermite=self._hermite)[0] in ("bdf_kernels.hip", "bdf_wave.hip")
        _native.build_code_object(self._source, constraints=self._constraints is not None, hermite=self._hermite,
                                  compact=self._compact)
        self._native = None
        self._last_forward = None
        # small batches run with more lanes per instance (measured: _native.small_batch_group); the code object of such a
        # mapping is built when the first batch of that size arrives
        self._batch_mapping = batch_mapping

    def _engine_kwargs(self):
        return dict(self._native_kwargs(), constraints=self._constraints, hermite=self._hermite,
                    compact=self._compact, guard_kinds=("adjoint",))

    def _set_tolerances(self, atol=None, rtol=None):
        atol, rtol = np.array(atol, dtype=float), np.array(rtol, dtype=float)
        if not (rtol.ndim == 0 and atol.ndim in (0, 1)):
            raise ValueError("Invalid tolerance.")
        if atol.ndim == 1 and atol.shape != (self._problem.n_states,):
            raise ValueError("Invalid tolerance.")
        self._atol, self._rtol = atol, rtol

    def _native_kwargs(self):
        rB, aB, rQ, aQ = self._tolB
        return dict(rtol=float(self._rtol), atol=self._atol, rtolB=rB, atolB=aB,
                    rtolQB=rQ, atolQB=aQ, mxstep=self._mxsteps, traj_capacity=self._max_steps,
                    arena_bytes=self._arena_bytes)

    def make_output_buffers(self, tvals):
        y_vals = np.zeros((len(tvals), self._problem.n_states))
        grad_out = np.zeros(self._problem.n_params)
        lamda_out = np.zeros(self._problem.n_states)
        return y_vals, grad_out, lamda_out

    # -- scalar API (B = 1) ----------------------------------------------------------------
    def solve_forward(self, t0, tvals, y0, y_out, *, max_retries=5):
        """``max_retries`` is accepted and forwarded like in the reference (solver.py:682,710-719); note that it can
        never take effect there either: CVodeF drives CVode in CV_ONE_STEP mode, whose per-call step counter
        never reaches mxstep, so CV_TOO_MUCH_WORK is not produced on this path."""
        y0 = _flat_state(self._problem, y0)
        ps, pr = self._problem.flat_params(self._user_data)
        yo, status, _ = self.solve_forward_batch(t0, tvals, y0[None], ps[None], pr, max_retries=max_retries)
        if status[0] != 0:
            code = int(status[0])
            if code == -1:
                raise SolverError("Too many solver retries.")
            if code == -9001:
                raise SolverError(f"Solving ode failed: more than max_steps={self._max_steps} stored forward steps "
                                  "(SA_STATUS_ARENA_FULL); raise max_steps / arena_gib")
            raise SolverError(f"Solving ode failed: {ERRORS.get(code, 'unknown')} ({code})")
        y_out[...] = yo[0]

    def solve_backward(self, t0, tend, tvals, grads, grad_out, lamda_out, lamda_all_out=None,
                       quad_all_out=None, max_retries=50):
        grads = np.ascontiguousarray(grads, dtype=np.float64)
        want_all = lamda_all_out is not None or quad_all_out is not None
        res = self.solve_backward_batch(t0, tend, tvals, grads[None], max_retries=max_retries, return_all=want_all)
        g, lam, status = res[0], res[1], res[2]
        if status[0] != 0:
            code = int(status[0])
            if code == -1:
                raise SolverError("Too many solver retries.")
            if code == -9001:
                raise SolverError(f"Solving ode failed: the stored forward trajectory does not fit the arena budget "
                                  f"or exceeds max_steps={self._max_steps} points (SA_STATUS_ARENA_FULL); raise "
                                  "arena_gib / max_steps")
            raise SolverError(f"Solving ode failed: {ERRORS.get(code, 'unknown')} ({code})")
        grad_out[...] = g[0]
        lamda_out[...] = lam[0]
        if lamda_all_out is not None:
            lamda_all_out[...] = res[4][0]
        if quad_all_out is not None:
            quad_all_out[...] = res[5][0]

    # -- batch API -------------------------------------------------------------------------
    def solve_forward_batch(self, t0, tvals, y0, params_sub, params_rem, *, max_retries=5, out=None):
        """B forward solves with stored trajectories: (y_out [B,n_t,n], status [B], stats [B,16]).

        ``out``: caller-allocated outputs written in place (the reference's convention,
        /root/reference/sunode/solver.py:682: ``solve_forward(t0, tvals, y0, y_out)``): a dict with some of
        ``y_out, status, stats`` or a sequence in return order; see ``Solver.solve_batch``."""
        self._select_mapping(np.shape(y0)[0] if np.ndim(y0) == 2 else 0)
        self._set_retries(max_retries_fwd=max_retries)
        B, y0, ps, pr, stride = self._batch_inputs(y0, params_sub, params_rem)
        tvals = np.ascontiguousarray(tvals, dtype=np.float64)
        p = self._problem.n_params
        y0, ps, pr = self._gather(y0), self._gather(ps, p), self._gather(pr, stride)
        names = ("y_out", "status", "stats")
        y_pair = self._out("fwd.y_out", (B, len(tvals), self._problem.n_states), given=self._out_arg(out, 0, names))
        st_pair = self._out("fwd.status", (B,), np.int32, self._out_arg(out, 1, names), small=True)
        sa_pair = self._out("fwd.stats", (B, _native.N_STATS), np.int64, self._out_arg(out, 2, names), small=True)
        y_out, status, stats = y_pair[0], st_pair[0], sa_pair[0]
        shards = self._shards(B)

        def call(eng, lo, hi):
            eng.solve(_native.SA_MEM_HOST, hi - lo, y0[lo:hi], _rows(ps, lo, hi, p), _rows(pr, lo, hi, stride),
                      stride, t0, tvals, len(tvals), y_out[lo:hi], status[lo:hi], stats[lo:hi], adjoint=True)
        self._run_shards(shards, call)
        self._last_forward = (B, ps, pr, stride, shards, self._mapping)     # every handle keeps ITS shard's trajectories
        return self._scatter(y_pair), self._scatter(st_pair), self._scatter(sa_pair)

    def solve_backward_batch(self, t0, tend, tvals, grads, *, max_retries=50, return_all=False, out=None):
        """Adjoint pass for the batch of the last ``solve_forward_batch``.

        ``grads``: [B, n_t, n] or [n_t, n] (shared).  Returns (grad_out [B,p] = dL/dp,
        lamda_out [B,n] = -dL/dy0, status [B], stats [B,16]); with ``return_all`` also
        (lamda_all [B,n_t,n], quad_all [B,n_t,p]): adjoint state and accumulated quadrature right
        after every jump, rows ordered as the reference's ``lamda_all_out[-i]`` (solver.py:778-781).
        ``out``: caller-allocated outputs written in place (reference: ``solve_backward(..., grad_out, lamda_out,
        lamda_all_out, quad_all_out)``, solver.py:723-724): a dict with some of ``grad_out [B,p], lamda_out [B,n],
        status, stats, lamda_all, quad_all`` or a sequence in return order."""
        if self._last_forward is None:
            raise SolverError("solve_backward called before solve_forward")
        B, ps, pr, stride, shards, self._mapping = self._last_forward      # (the handles that integrated forward)
        self._set_retries(max_retries_bwd=max_retries)
        n, p = self._problem.n_states, self._problem.n_params
        tvals = np.ascontiguousarray(tvals, dtype=np.float64)
        n_t = len(tvals)
        grads = np.ascontiguousarray(grads, dtype=np.float64)
        if grads.shape == (n_t, n):
            gstride = 0
        elif grads.shape == (B, n_t, n):
            gstride = n_t * n
        else:
            raise ValueError(f"grads must have shape ({n_t}, {n}) or ({B}, {n_t}, {n})")
        grads = self._gather(grads, gstride)                 # (ps / pr of the forward call are in handle order already)
        names = ("grad_out", "lamda_out", "status", "stats", "lamda_all", "quad_all")
        if out is not None and not return_all and (len(out) > 4 if not isinstance(out, dict)
                                                   else {"lamda_all", "quad_all"} & set(out)):
            raise ValueError("out= names lamda_all / quad_all but return_all is False")
        # (the library wants at least one column: with p == 0 / n == 0 it writes into a private dummy)
        g_pair = self._out("bwd.grad_out", (B, p), given=self._out_arg(out, 0, names)) if p else None
        l_pair = self._out("bwd.lamda_out", (B, n), given=self._out_arg(out, 1, names)) if n else None
        grad_out = g_pair[0] if p else np.zeros((B, 1))
        lamda_out = l_pair[0] if n else np.zeros((B, 1))
        st_pair = self._out("bwd.status", (B,), np.int32, self._out_arg(out, 2, names), small=True)
        sa_pair = self._out("bwd.stats", (B, _native.N_STATS), np.int64, self._out_arg(out, 3, names), small=True)
        status, stats = st_pair[0], sa_pair[0]
        la_pair = qa_pair = None
        lam_all = quad_all = None
        if return_all:
            la_pair = self._out("bwd.lamda_all", (B, n_t, n), given=self._out_arg(out, 4, names)) if n else None
            qa_pair = self._out("bwd.quad_all", (B, n_t, p), given=self._out_arg(out, 5, names)) if p else None
            lam_all = la_pair[0] if n else np.zeros((B, n_t, 1))
            quad_all = qa_pair[0] if p else np.zeros((B, n_t, 1))

        def call(eng, lo, hi):
            eng.solve_backward(_native.SA_MEM_HOST, hi - lo, _rows(ps, lo, hi, p), _rows(pr, lo, hi, stride), stride,
                               t0, tend, tvals, n_t, _rows(grads, lo, hi, gstride), gstride, grad_out[lo:hi],
                               lamda_out[lo:hi], status[lo:hi], stats[lo:hi],
                               lam_all[lo:hi] if return_all else None, quad_all[lo:hi] if return_all else None)
        self._run_shards(shards, call)
        # (p == 0 / n == 0: the caller's (B, 0) array, if any, has nothing to receive)
        grad_out = self._scatter(g_pair) if p else self._zero_width(self._out_arg(out, 0, names), (B, 0))
        lamda_out = self._scatter(l_pair) if n else self._zero_width(self._out_arg(out, 1, names), (B, 0))
        status, stats = self._scatter(st_pair), self._scatter(sa_pair)
        if return_all:
            lam_all = self._scatter(la_pair) if n else self._zero_width(self._out_arg(out, 4, names), (B, n_t, 0))
            quad_all = self._scatter(qa_pair) if p else self._zero_width(self._out_arg(out, 5, names), (B, n_t, 0))
        if (status == -9001).any():
            import warnings
            warnings.warn("solve_backward: %d instance(s) returned SA_STATUS_ARENA_FULL (a 64-instance group of stored "
                          "trajectories exceeds the arena budget, or an instance more than max_steps=%d points); their "
                          "gradients are NaN -- raise AdjointSolver(arena_gib=..., max_steps=...)"
                          % (int((status == -9001).sum()), self._max_steps), RuntimeWarning, stacklevel=2)
        if return_all:
            return grad_out, lamda_out, status, stats, lam_all, quad_all
        return grad_out, lamda_out, status, stats
