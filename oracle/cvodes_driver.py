"""Real CVODES through ctypes, driven exactly like the reference drives it -- when the host has it.

BASELINE.md 3.2 / SURVEY.md 8(d): the reference's CPU path is sunode + libsundials_cvodes (conda `sundials<6`), which
cannot travel to the GPU box.  If `ctypes.util.find_library("sundials_cvodes")` succeeds there, this module times the
real thing: the call sequence of AdjointSolver.__init__ / _init_backward / solve_forward / solve_backward
(/root/reference/sunode/solver.py:565-615, 682-784), callbacks compiled from the generated C header
(oracle/cvodes_callbacks.c) instead of numba.  It is also the only way this repository can be pinned against CVODES
itself: `compare_with_oracle` reports states / gradients / step counters next to the restated oracle's.
TEST / BASELINE INFRASTRUCTURE: used by bench.py's cpu_baseline leg and tests only.  The call order is unit-tested
against a recording fake (tests/test_cvodes_driver.py); the arithmetic needs the library.
"""
import ctypes
import ctypes.util
import os
import subprocess

import numpy as np

CV_BDF, CV_NORMAL, CV_POLYNOMIAL, CV_TOO_MUCH_WORK = 2, 1, 2, -1
LIBS = ("sundials_nvecserial", "sundials_sunmatrixdense", "sundials_sunlinsoldense", "sundials_cvodes")
_dbl, _vp, _int, _long = ctypes.c_double, ctypes.c_void_p, ctypes.c_int, ctypes.c_long


def find_libraries():
    """{name: path} of the four SUNDIALS libraries the reference links (build_cvodes.py:63-71), or None."""
    found = {n: ctypes.util.find_library(n) for n in LIBS}
    return found if all(found.values()) else None


def build_callbacks(native_source: str, tag: str) -> str:
    """gcc -O2 build of oracle/cvodes_callbacks.c around the problem's generated header."""
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "_build")
    os.makedirs(out, exist_ok=True)
    hdr, lib = os.path.join(out, "cvcb_%s.h" % tag), os.path.join(out, "cvcb_%s.so" % tag)
    with open(hdr, "w") as fh:
        fh.write(native_source)
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-DSA_PROBLEM_HEADER=\"%s\"" % hdr,
                    os.path.join(here, "cvodes_callbacks.c"), "-o", lib, "-lm"], check=True)
    return lib


class _UserData(ctypes.Structure):
    _fields_ = [("ps", ctypes.POINTER(_dbl)), ("pr", ctypes.POINTER(_dbl))]


class CvodesDriver:
    """One CVODES forward + backward problem (one draw at a time, as the reference)."""

    def __init__(self, n, p, cb, sun, *, rtol, atol, rtolB, atolB, rtolQB, atolQB, checkpoint_n=500_000):
        """cb: CDLL of build_callbacks(); sun: object with the SUNDIALS entry points as attributes (one CDLL per
        library merged by `load`, or the recording fake of the unit test)."""
        self.n, self.p, self.S, self.cb = n, p, sun, cb
        S = sun
        for name in ("N_VNew_Serial", "SUNDenseMatrix", "SUNLinSol_Dense", "CVodeCreate", "N_VGetArrayPointer",
                     "SUNDenseMatrix_Data"):
            getattr(S, name).restype = _vp
        cb_ptr = lambda f: ctypes.cast(f, _vp)                                              # noqa: E731
        ctypes.c_void_p.in_dll(cb, "sa_nv_data").value = ctypes.cast(S.N_VGetArrayPointer, _vp).value
        ctypes.c_void_p.in_dll(cb, "sa_dm_data").value = ctypes.cast(S.SUNDenseMatrix_Data, _vp).value
        self.ud = _UserData()
        udp = ctypes.cast(ctypes.pointer(self.ud), _vp)
        vec = lambda k: _vp(S.N_VNew_Serial(_long(max(k, 1))))                              # noqa: E731
        self.y, self.yB, self.q, self.qout = vec(n), vec(n), vec(p), vec(p)
        self.J, self.JB = _vp(S.SUNDenseMatrix(_long(n), _long(n))), _vp(S.SUNDenseMatrix(_long(n), _long(n)))
        # AdjointSolver.__init__ (solver.py:565-586)
        self.ode = _vp(S.CVodeCreate(CV_BDF))
        self._ok(S.CVodeInit(self.ode, cb_ptr(cb.sa_cv_rhs), _dbl(0.0), self.y))
        self._ok(S.CVodeSStolerances(self.ode, _dbl(rtol), _dbl(atol)))
        self._ok(S.CVodeSetLinearSolver(self.ode, _vp(S.SUNLinSol_Dense(self.y, self.J)), self.J))
        self._ok(S.CVodeSetJacFn(self.ode, cb_ptr(cb.sa_cv_jac)))
        self._ok(S.CVodeSetUserData(self.ode, udp))
        # _init_backward (solver.py:588-615)
        self._ok(S.CVodeAdjInit(self.ode, _long(checkpoint_n), CV_POLYNOMIAL))
        which = _int(0)
        self._ok(S.CVodeCreateB(self.ode, CV_BDF, ctypes.byref(which)))
        self.B = which.value
        self._ok(S.CVodeInitB(self.ode, self.B, cb_ptr(cb.sa_cv_rhsB), _dbl(0.0), self.yB))
        self._ok(S.CVodeSStolerancesB(self.ode, self.B, _dbl(rtolB), _dbl(atolB)))
        self._ok(S.CVodeSetLinearSolverB(self.ode, self.B, _vp(S.SUNLinSol_Dense(self.yB, self.JB)), self.JB))
        self._ok(S.CVodeSetJacFnB(self.ode, self.B, cb_ptr(cb.sa_cv_jacB)))
        self._ok(S.CVodeSetUserDataB(self.ode, self.B, udp))
        self._ok(S.CVodeQuadInitB(self.ode, self.B, cb_ptr(cb.sa_cv_quadB), self.q))
        self._ok(S.CVodeQuadSStolerancesB(self.ode, self.B, _dbl(rtolQB), _dbl(atolQB)))
        self._ok(S.CVodeSetQuadErrConB(self.ode, self.B, 1))

    @staticmethod
    def _ok(rc):
        if rc != 0:
            raise RuntimeError("CVODES call failed: %d" % rc)

    def _view(self, v, k):
        return np.ctypeslib.as_array(ctypes.cast(_vp(self.S.N_VGetArrayPointer(v)), ctypes.POINTER(_dbl)), (max(k, 1),))[:k]

    def set_params(self, ps, pr):
        self._ps, self._pr = np.ascontiguousarray(np.r_[ps, 0.0]), np.ascontiguousarray(np.r_[pr, 0.0])
        self.ud.ps = self._ps.ctypes.data_as(ctypes.POINTER(_dbl))
        self.ud.pr = self._pr.ctypes.data_as(ctypes.POINTER(_dbl))

    def solve_forward(self, t0, tvals, y0, max_retries=5):          # solver.py:682-721
        S, tret, ncheck = self.S, _dbl(t0), _int(0)
        yv = self._view(self.y, self.n)
        yv[:] = y0
        self._ok(S.CVodeReInit(self.ode, _dbl(t0), self.y))
        self._ok(S.CVodeAdjReInit(self.ode))
        out = np.zeros((len(tvals), self.n))
        for i, t in enumerate(tvals):
            if t == t0:
                out[0] = y0
                continue
            for _ in range(max_retries):
                rc = S.CVodeF(self.ode, _dbl(t), self.y, ctypes.byref(tret), CV_NORMAL, ctypes.byref(ncheck))
                if rc == 0:
                    break
                if rc != CV_TOO_MUCH_WORK:
                    raise RuntimeError("CVodeF failed: %d" % rc)
            else:
                raise RuntimeError("Too many solver retries.")
            out[i] = yv
        return out

    def solve_backward(self, t0, tend, tvals, grads, max_retries=50):    # solver.py:723-784
        S, tret = self.S, _dbl(t0)
        lam, quad, qout = self._view(self.yB, self.n), self._view(self.q, self.p), self._view(self.qout, self.p)
        lam[:] = 0; quad[:] = 0; qout[:] = 0
        ts = [t0] + list(tvals[::-1]) + [tend]
        for (t_lower, t_upper), g in zip(zip(ts[1:], ts[:-1]), list(grads[::-1]) + [None]):
            if t_lower < t_upper:
                self._ok(S.CVodeReInitB(self.ode, self.B, _dbl(t_upper), self.yB))
                self._ok(S.CVodeQuadReInitB(self.ode, self.B, self.q))
                for _ in range(max_retries):
                    rc = S.CVodeB(self.ode, _dbl(t_lower), CV_NORMAL)
                    if rc == 0:
                        break
                    if rc != CV_TOO_MUCH_WORK:
                        raise RuntimeError("CVodeB failed: %d" % rc)
                else:
                    raise RuntimeError("Too many solver retries.")
                self._ok(S.CVodeGetB(self.ode, self.B, ctypes.byref(tret), self.yB))
                self._ok(S.CVodeGetQuadB(self.ode, self.B, ctypes.byref(tret), self.qout))
                quad[:] = qout
            if g is not None:
                lam -= g
        return qout.copy(), lam.copy()

    def counters(self):
        """nst, nfe, nsetups, nje, nni, ncfn, netf of the forward problem (CVodeGetNumSteps & friends)."""
        out = []
        for name in ("CVodeGetNumSteps", "CVodeGetNumRhsEvals", "CVodeGetNumLinSolvSetups", "CVodeGetNumJacEvals",
                     "CVodeGetNumNonlinSolvIters", "CVodeGetNumNonlinSolvConvFails", "CVodeGetNumErrTestFails"):
            v = _long(0)
            self._ok(getattr(self.S, name)(self.ode, ctypes.byref(v)))
            out.append(v.value)
        return out


class _Merged:
    """Attribute lookup over several CDLLs (the SUNDIALS entry points live in four libraries)."""

    def __init__(self, libs):
        self._libs = libs

    def __getattr__(self, name):
        for L in self._libs:
            try:
                return getattr(L, name)
            except AttributeError:
                continue
        raise AttributeError(name)


def load(paths):
    return _Merged([ctypes.CDLL(paths[n], mode=ctypes.RTLD_GLOBAL) for n in LIBS])
