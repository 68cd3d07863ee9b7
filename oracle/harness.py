"""ctypes harness around oracle/cvodes_oracle.c -- TEST INFRASTRUCTURE ONLY.

Builds one shared object per problem (the generated callback header is compiled
into the integrator, the way CVODES links against the numba cfuncs in the
reference) under ``oracle/_build/`` and exposes batch drivers that mirror
``Solver.solve`` / ``AdjointSolver.solve_forward`` / ``solve_backward``
(/root/reference/sunode/solver.py:467-527, 682-784).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this module.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_SRC = os.path.join(_HERE, "cvodes_oracle.c")

N_STATS = 16
STAT_NAMES = ["nst", "nfe", "nsetups", "nje", "nni", "ncfn", "netf", "qlast",
              "npts", "nfqe", "netfq", "ninterp", "nrebuild", "retries", "r0", "r1"]

_dp = ctypes.POINTER(ctypes.c_double)


def _ptr(a: np.ndarray, ctype=ctypes.c_double):
    return a.ctypes.data_as(ctypes.POINTER(ctype))


def build_oracle_library(native_source: str, tag: Optional[str] = None, opt: str = "-O2") -> str:
    """Compile cvodes_oracle.c against one generated problem header; returns the .so path."""
    os.makedirs(_BUILD, exist_ok=True)
    with open(_SRC, "rb") as fh:
        src = fh.read()
    fma_flag = []
    try:
        with open("/proc/cpuinfo") as fh:
            if " fma " in fh.read():
                fma_flag = ["-mfma"]          # __builtin_fma -> vfmadd instead of a libm call
    except OSError:
        pass
    key = hashlib.sha256(native_source.encode() + b"\0" + src + opt.encode() +
                         " ".join(fma_flag).encode()).hexdigest()[:16]
    stem = "orc_%s_%s" % (tag or "p", key)
    hdr = os.path.join(_BUILD, stem + ".h")
    lib = os.path.join(_BUILD, stem + ".so")
    if not os.path.exists(lib):
        with open(hdr, "w") as fh:
            fh.write(native_source)
        cmd = ["gcc", opt, "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-std=gnu11"] + fma_flag + [
               "-DSA_PROBLEM_HEADER=\"%s\"" % hdr, _SRC, "-o", lib + ".tmp", "-lm"]
        subprocess.run(cmd, check=True, capture_output=True, text=True)
        os.replace(lib + ".tmp", lib)
    return lib


class OracleConfig:
    def __init__(self, n_states, rtol=1e-10, atol=1e-10, rtolB=1e-10, atolB=1e-10,
                 rtolQB=1e-10, atolQB=1e-10, mxstep=500, max_retries_fwd=5,
                 max_retries_bwd=50, max_traj_points=0, constraints=None, hermite=False, errconQB=True):
        nsd = max(n_states, 1)

        class _Cfg(ctypes.Structure):
            _fields_ = [("rtol", ctypes.c_double), ("atol", ctypes.c_double * nsd),
                        ("rtolB", ctypes.c_double), ("atolB", ctypes.c_double),
                        ("rtolQB", ctypes.c_double), ("atolQB", ctypes.c_double),
                        ("mxstep", ctypes.c_int), ("max_retries_fwd", ctypes.c_int),
                        ("max_retries_bwd", ctypes.c_int), ("max_traj_points", ctypes.c_int),
                        ("constraints_set", ctypes.c_int), ("hermite", ctypes.c_int),
                        ("constraints", ctypes.c_double * nsd), ("no_errconQB", ctypes.c_int),
                        ("reserved_pad", ctypes.c_int)]
        c = _Cfg()
        c.rtol = rtol
        av = np.broadcast_to(np.asarray(atol, dtype=float), (n_states,)) if n_states else []
        for i, v in enumerate(av):
            c.atol[i] = float(v)
        c.rtolB, c.atolB, c.rtolQB, c.atolQB = rtolB, atolB, rtolQB, atolQB
        c.mxstep, c.max_retries_fwd, c.max_retries_bwd = mxstep, max_retries_fwd, max_retries_bwd
        c.max_traj_points = max_traj_points
        c.constraints_set = 0 if constraints is None else 1
        c.hermite = 1 if hermite else 0
        c.no_errconQB = 0 if errconQB else 1
        if constraints is not None:
            for i, v in enumerate(np.broadcast_to(np.asarray(constraints, dtype=float), (n_states,))):
                c.constraints[i] = float(v)
        self.c = c


class Oracle:
    """Batch front-end of the CPU oracle for one problem."""

    def __init__(self, problem, tag: Optional[str] = None, opt: str = "-O2"):
        self.problem = problem
        self.lib_path = build_oracle_library(problem.native_source(), tag, opt)
        L = ctypes.CDLL(self.lib_path)
        self.L = L
        n, p, r, k = (ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int())
        L.orc_sizes(ctypes.byref(n), ctypes.byref(p), ctypes.byref(r), ctypes.byref(k))
        self.n, self.p, self.r = n.value, p.value, r.value
        assert k.value == N_STATS
        assert (self.n, self.p) == (problem.n_states, problem.n_params)
        L.orc_batch_new.restype = ctypes.c_void_p
        L.orc_batch_free.argtypes = [ctypes.c_void_p]
        L.orc_det_pow.restype = ctypes.c_double
        L.orc_det_pow.argtypes = [ctypes.c_double, ctypes.c_double]
        self._batch = None
        self._batch_B = 0

    def __del__(self):
        try:
            if self._batch:
                self.L.orc_batch_free(self._batch)
        except Exception:
            pass

    def config(self, **kw) -> OracleConfig:
        cfg = OracleConfig(self.n, **kw)
        assert ctypes.sizeof(cfg.c) == self.L.orc_config_size()
        return cfg

    def det_pow(self, x: float, y: float) -> float:
        return self.L.orc_det_pow(x, y)

    # -- callbacks ---------------------------------------------------------
    def eval(self, t, y, lam, ps, pr):
        n, p = self.n, self.p
        y = np.ascontiguousarray(y, float); lam = np.ascontiguousarray(lam, float)
        pr = self._extend(pr)
        ps = np.ascontiguousarray(np.r_[ps, 0.0], float); pr = np.ascontiguousarray(np.r_[pr, 0.0], float)
        rhs = np.zeros(max(n, 1)); jac = np.zeros(max(n * n, 1)); adj = np.zeros(max(n, 1))
        quad = np.zeros(max(p, 1)); adjjac = np.zeros(max(n * n, 1))
        codes = np.zeros(5, np.int32)
        self.L.orc_eval(ctypes.c_double(t), _ptr(y), _ptr(lam), _ptr(ps), _ptr(pr), _ptr(rhs), _ptr(jac),
                        _ptr(adj), _ptr(quad), _ptr(adjjac), _ptr(codes, ctypes.c_int))
        return dict(rhs=rhs[:n], jac=jac[:n * n].reshape(n, n).T.copy(), adj=adj[:n], quad=quad[:p],
                    adjjac=adjjac[:n * n].reshape(n, n).T.copy(), codes=codes)

    # -- helpers -----------------------------------------------------------
    def _extend(self, pr):
        """user remainder vector -> the one the generated source reads (hoisted fixed-parameter
        sub-expressions appended; see SympyProblem.extend_remainder)"""
        pr = np.asarray(pr, float)
        if pr.size and pr.shape[-1] == self.problem.n_remainder and self.r != self.problem.n_remainder:
            pr = self.problem.extend_remainder(pr)
        return pr

    def _params(self, B, ps, pr):
        ps = np.ascontiguousarray(np.broadcast_to(np.asarray(ps, float).reshape(-1, self.p) if self.p else
                                                  np.zeros((B, 0)), (B, self.p)))
        pr = self._extend(pr)
        if pr.ndim == 1 or (pr.ndim == 2 and pr.shape[0] == 1 and B != 1):
            pr2 = np.ascontiguousarray(pr.reshape(-1)); stride = 0
        else:
            pr2 = np.ascontiguousarray(pr.reshape(B, self.r)); stride = self.r
        # keep pointers valid for empty arrays
        if ps.size == 0:
            ps = np.zeros((B, 1))
        if pr2.size == 0:
            pr2 = np.zeros(1)
        return ps, pr2, stride

    def _ensure_batch(self, B):
        if self._batch is None or self._batch_B < B:
            if self._batch:
                self.L.orc_batch_free(self._batch)
            self._batch = self.L.orc_batch_new(B)
            self._batch_B = B

    # -- drivers -------------------------------------------------------------
    def solve(self, cfg, y0, ps, pr, t0, tvals, nthreads=1):
        y0 = np.ascontiguousarray(np.asarray(y0, float).reshape(-1, self.n)); B = y0.shape[0]
        ps, pr, stride = self._params(B, ps, pr)
        tvals = np.ascontiguousarray(tvals, float); n_t = len(tvals)
        y_out = np.zeros((B, n_t, self.n)); status = np.zeros(B, np.int32); stats = np.zeros((B, N_STATS), np.int64)
        self.L.orc_solve_batch(ctypes.byref(cfg.c), B, _ptr(y0), _ptr(ps), _ptr(pr), stride,
                               ctypes.c_double(t0), _ptr(tvals), n_t, _ptr(y_out),
                               _ptr(status, ctypes.c_int32), _ptr(stats, ctypes.c_int64), nthreads)
        return y_out, status, stats

    def solve_sens(self, cfg, y0, ps, pr, sens0, t0, tvals, mode="simultaneous", scaling_factors=None, nthreads=1):
        """Solver(sens_mode=...).solve: returns (y_out [B,n_t,n], sens_out [B,n_t,p,n], status, stats)."""
        y0 = np.ascontiguousarray(np.asarray(y0, float).reshape(-1, self.n)); B = y0.shape[0]
        ps, pr, stride = self._params(B, ps, pr)
        sens0 = np.ascontiguousarray(np.broadcast_to(np.asarray(sens0, float), (B, self.p, self.n)))
        tvals = np.ascontiguousarray(tvals, float); n_t = len(tvals)
        y_out = np.zeros((B, n_t, self.n)); sens_out = np.zeros((B, n_t, self.p, self.n))
        status = np.zeros(B, np.int32); stats = np.zeros((B, N_STATS), np.int64)
        pbar = None if scaling_factors is None else np.ascontiguousarray(scaling_factors, float)
        self.L.orc_solve_sens_batch(ctypes.byref(cfg.c), {"simultaneous": 0, "staggered": 1}[mode],
                                    _ptr(pbar) if pbar is not None else None, B, _ptr(y0), _ptr(ps), _ptr(pr),
                                    stride, _ptr(sens0), ctypes.c_double(t0), _ptr(tvals), n_t, _ptr(y_out),
                                    _ptr(sens_out), _ptr(status, ctypes.c_int32), _ptr(stats, ctypes.c_int64),
                                    nthreads)
        return y_out, sens_out, status, stats

    def solve_forward(self, cfg, y0, ps, pr, t0, tvals, nthreads=1):
        y0 = np.ascontiguousarray(np.asarray(y0, float).reshape(-1, self.n)); B = y0.shape[0]
        ps, pr, stride = self._params(B, ps, pr)
        self._ensure_batch(B)
        self._fwd = (B, ps, pr, stride)
        tvals = np.ascontiguousarray(tvals, float); n_t = len(tvals)
        y_out = np.zeros((B, n_t, self.n)); status = np.zeros(B, np.int32); stats = np.zeros((B, N_STATS), np.int64)
        self.L.orc_solve_forward_batch(ctypes.c_void_p(self._batch), ctypes.byref(cfg.c), B, _ptr(y0), _ptr(ps),
                                       _ptr(pr), stride, ctypes.c_double(t0), _ptr(tvals), n_t, _ptr(y_out),
                                       _ptr(status, ctypes.c_int32), _ptr(stats, ctypes.c_int64), nthreads)
        return y_out, status, stats

    def solve_backward(self, cfg, t0, tend, tvals, grads, nthreads=1, return_all=False):
        B, ps, pr, stride = self._fwd
        tvals = np.ascontiguousarray(tvals, float); n_t = len(tvals)
        grads = np.ascontiguousarray(grads, float)
        gstride = 0 if grads.ndim == 2 else n_t * self.n
        grad_out = np.zeros((B, max(self.p, 1))); lamda_out = np.zeros((B, max(self.n, 1)))
        status = np.zeros(B, np.int32); stats = np.zeros((B, N_STATS), np.int64)
        lam_all = np.zeros((B, n_t, max(self.n, 1))); quad_all = np.zeros((B, n_t, max(self.p, 1)))
        self.L.orc_solve_backward_batch(ctypes.c_void_p(self._batch), ctypes.byref(cfg.c), B, _ptr(ps), _ptr(pr),
                                        stride, ctypes.c_double(t0), ctypes.c_double(tend), _ptr(tvals), n_t,
                                        _ptr(grads), ctypes.c_long(gstride), _ptr(grad_out), _ptr(lamda_out),
                                        _ptr(status, ctypes.c_int32), _ptr(stats, ctypes.c_int64), nthreads,
                                        _ptr(lam_all) if return_all else None, _ptr(quad_all) if return_all else None)
        if return_all:
            return grad_out[:, :self.p], lamda_out[:, :self.n], status, stats, lam_all, quad_all[:, :, :self.p]
        return grad_out[:, :self.p], lamda_out[:, :self.n], status, stats

    def trajectory(self, i):
        n = self.L.orc_traj_len(ctypes.c_void_p(self._batch), i)
        t = np.zeros(max(n, 1)); y = np.zeros((max(n, 1), max(self.n, 1))); o = np.zeros(max(n, 1), np.int32)
        self.L.orc_traj_get(ctypes.c_void_p(self._batch), i, _ptr(t), _ptr(y), _ptr(o, ctypes.c_int))
        return t[:n], y[:n, :self.n], o[:n]
