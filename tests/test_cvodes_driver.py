"""oracle/cvodes_driver.py: the real-CVODES leg of the CPU baseline cannot run in this image (no SUNDIALS), so its
CALL ORDER is checked against a recording fake: it must be the reference's sequence
(/root/reference/sunode/solver.py:565-615 set-up, :682-721 solve_forward, :723-784 solve_backward)."""
import ctypes

import numpy as np

from oracle import cvodes_driver as drv
from tests.helpers import make_problem


class FakeSundials:
    """Records every call; vectors / matrices are real numpy buffers so the driver's views work."""

    def __init__(self):
        self.calls, self._keep = [], []
        self._proto = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p)
        self.N_VGetArrayPointer = self._proto(lambda v: v)          # a vector handle IS its data pointer
        self.SUNDenseMatrix_Data = self._proto(lambda v: v)

    def _buf(self, k):
        a = np.zeros(max(int(k), 1))
        self._keep.append(a)
        return a.ctypes.data

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        rec = self.calls

        class F:
            restype = None

            def __call__(self_, *args):
                rec.append(name)
                if name == "N_VNew_Serial":
                    return self._buf(args[0].value)
                if name == "SUNDenseMatrix":
                    return self._buf(args[0].value * args[1].value)
                if name in ("SUNLinSol_Dense", "CVodeCreate"):
                    return self._buf(1)
                if name == "CVodeCreateB":
                    args[2]._obj.value = 0
                return 0
        f = F()
        object.__setattr__(self, name, f)
        return f


def test_call_sequence_is_the_reference_sequence(tmp_path):
    prob = make_problem("lv")
    cb = ctypes.CDLL(drv.build_callbacks(prob.native_source(), "lv_test"))
    S = FakeSundials()
    d = drv.CvodesDriver(2, 2, cb, S, rtol=1e-8, atol=1e-8, rtolB=1e-8, atolB=1e-8, rtolQB=1e-8, atolQB=1e-8)
    setup = [c for c in S.calls if c not in ("N_VNew_Serial", "SUNDenseMatrix")]
    assert setup == ["CVodeCreate", "CVodeInit", "CVodeSStolerances", "SUNLinSol_Dense", "CVodeSetLinearSolver",
                     "CVodeSetJacFn", "CVodeSetUserData", "CVodeAdjInit", "CVodeCreateB", "CVodeInitB",
                     "CVodeSStolerancesB", "SUNLinSol_Dense", "CVodeSetLinearSolverB", "CVodeSetJacFnB",
                     "CVodeSetUserDataB", "CVodeQuadInitB", "CVodeQuadSStolerancesB", "CVodeSetQuadErrConB"]
    d.set_params(np.array([0.1, 0.2]), np.array([0.3, 0.4]))
    del S.calls[:]
    tv = np.array([0.0, 1.0, 2.0])
    y = d.solve_forward(0.0, tv, np.array([1.0, 0.1]))
    assert S.calls == ["CVodeReInit", "CVodeAdjReInit", "CVodeF", "CVodeF"] and y.shape == (3, 2)
    np.testing.assert_array_equal(y[0], [1.0, 0.1])
    del S.calls[:]
    g, lam = d.solve_backward(2.0, 0.0, tv, np.ones((3, 2)))
    per_interval = ["CVodeReInitB", "CVodeQuadReInitB", "CVodeB", "CVodeGetB", "CVodeGetQuadB"]
    assert S.calls == per_interval * 2            # (2 -> 2) is empty, (2 -> 1), (1 -> 0), (0 -> 0) is empty
    np.testing.assert_array_equal(lam, [-3.0, -3.0])    # the fake integrates nothing: three jumps of -1


def test_generated_callbacks_have_the_cvodes_abi():
    """The shim compiles around a generated header and its five entry points compute what the oracle's do."""
    prob = make_problem("lv")
    cb = ctypes.CDLL(drv.build_callbacks(prob.native_source(), "lv_test"))
    ident = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p)(lambda v: v)
    for sym in ("sa_nv_data", "sa_dm_data"):
        ctypes.c_void_p.in_dll(cb, sym).value = ctypes.cast(ident, ctypes.c_void_p).value
    ps, pr = np.array([0.1, 0.2, 0.0]), np.array([0.3, 0.4, 0.0])
    ud = drv._UserData(ps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), pr.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    y, out, J = np.array([1.0, 0.1]), np.zeros(2), np.zeros(4)
    vp = ctypes.c_void_p
    rc = cb.sa_cv_rhs(ctypes.c_double(0.0), vp(y.ctypes.data), vp(out.ctypes.data), ctypes.byref(ud))
    np.testing.assert_allclose(out, [0.08, 0.01], rtol=1e-14)     # README LV at y=(1,0.1)
    assert rc == 0
    cb.sa_cv_jac(ctypes.c_double(0.0), vp(y.ctypes.data), None, vp(J.ctypes.data), ctypes.byref(ud), None, None, None)
    np.testing.assert_allclose(J.reshape(2, 2).T, [[0.08, -0.2], [0.04, 0.1]], rtol=1e-14)
