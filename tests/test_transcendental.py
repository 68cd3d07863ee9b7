"""Transcendental right-hand sides and the reference's symbolic helper functions -- the CPU half.

* ``csrc/sa_math.h`` (the deterministic exp / log / sin / pow ... every generated header embeds): accuracy against
  mpmath, special values;
* ``sunode.symode.lambdify`` (alias of ``sunode_amd.symode.lambdify``): the reference's helper classes
  (/root/reference/sunode/symode/lambdify.py:275-352) with working derivative rules, checked against sympy's own
  differentiation of the exp / piecewise forms;
* the oracle (host build of the generated callbacks + the restated integrator) on ``forcing`` / ``logistic_switch`` /
  ``misc`` against DOP853 truth (tools/make_golden_truth.py --transcendental).

The GPU half (device == oracle bit for bit, device vs truth) is tests/test_gpu_transcendental.py.
"""
import ctypes
import hashlib
import os
import subprocess

import numpy as np
import pytest

from tests.helpers import make_oracle, make_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ONE = ["exp", "expm1", "log", "log1p", "sin", "cos", "tan", "tanh", "sinh", "cosh", "expit", "dexpit",
       "cardinal_bspline4"]
TWO = ["pow", "logaddexp"]


@pytest.fixture(scope="module")
def mathlib():
    """sa_math.h compiled for the host exactly like the oracle compiles a generated header."""
    hdr = os.path.join(ROOT, "sunode_amd", "csrc", "sa_math.h")
    with open(hdr, "rb") as fh:
        key = hashlib.sha256(fh.read()).hexdigest()[:12]
    out = os.path.join(ROOT, "oracle", "_build", "sa_math_%s.so" % key)
    if not os.path.exists(out):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        src = ["#include <math.h>", "#define SA_FN static inline", '#include "%s"' % hdr]
        src += ["void w_%s(int n, const double *x, double *o) { for (int i = 0; i < n; i++) o[i] = sa_%s(x[i]); }"
                % (f, f) for f in ONE]
        src += ["void w_%s(int n, const double *x, const double *y, double *o) "
                "{ for (int i = 0; i < n; i++) o[i] = sa_%s(x[i], y[i]); }" % (f, f) for f in TWO]
        c = out[:-3] + ".c"
        with open(c, "w") as fh:
            fh.write("\n".join(src) + "\n")
        with open("/proc/cpuinfo") as fh:
            fma = ["-mfma"] if " fma " in fh.read() else []
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-std=gnu11"] + fma +
                       [c, "-o", out, "-lm"], check=True, capture_output=True, text=True)
    L = ctypes.CDLL(out)

    def call(name, x, y=None):
        x = np.ascontiguousarray(x, float)
        o = np.empty_like(x)
        vp = ctypes.c_void_p
        if y is None:
            getattr(L, "w_" + name)(len(x), vp(x.ctypes.data), vp(o.ctypes.data))
        else:
            y = np.ascontiguousarray(y, float)
            getattr(L, "w_" + name)(len(x), vp(x.ctypes.data), vp(y.ctypes.data), vp(o.ctypes.data))
        return o
    return call


def _ulps(got, ref):
    import mpmath as mp
    worst = 0.0
    for g, r in zip(got, ref):
        rf = float(r)
        if not np.isfinite(rf) or rf == 0.0:
            assert g == rf or (np.isnan(g) and np.isnan(rf)), (g, rf)
            continue
        worst = max(worst, float(abs(mp.mpf(float(g)) - r) / np.spacing(abs(rf))))
    return worst


def test_sa_math_accuracy_against_mpmath(mathlib):
    """The bounds sa_math.h states (exp / log / sin / cos < 1 ulp, the others <= 4)."""
    import mpmath as mp
    mp.mp.prec = 200
    rng = np.random.RandomState(0)
    N = 1500
    sgn = rng.choice([-1.0, 1.0], N)
    cases = {
        "exp": (np.concatenate([rng.uniform(-745, 709.7, N), rng.uniform(-2, 2, N)]), mp.exp, 1.0),
        "expm1": (np.concatenate([rng.uniform(-40, 40, N), rng.uniform(-1, 1, N), 10.0 ** rng.uniform(-20, 0, N) * sgn]),
                  mp.expm1, 2.5),
        "log": (np.concatenate([10.0 ** rng.uniform(-320, 308, N), rng.uniform(0.5, 2, N)]), mp.log, 1.0),
        "log1p": (np.concatenate([rng.uniform(-0.999, 5, N), 10.0 ** rng.uniform(-20, 0, N) * sgn,
                                  10.0 ** rng.uniform(0, 300, N)]), mp.log1p, 2.5),
        "sin": (np.concatenate([rng.uniform(-10, 10, N), rng.uniform(-1e6, 1e6, N), 10.0 ** rng.uniform(-10, 15, N)]),
                mp.sin, 1.0),
        "cos": (np.concatenate([rng.uniform(-10, 10, N), rng.uniform(-1e6, 1e6, N), 10.0 ** rng.uniform(-10, 15, N)]),
                mp.cos, 1.0),
        "tan": (np.concatenate([rng.uniform(-10, 10, N), rng.uniform(-1e6, 1e6, N)]), mp.tan, 3.0),
        "tanh": (np.concatenate([rng.uniform(-25, 25, N), 10.0 ** rng.uniform(-20, 0, N)]), mp.tanh, 4.0),
        "sinh": (np.concatenate([rng.uniform(-710, 710, N), 10.0 ** rng.uniform(-20, 1, N)]), mp.sinh, 3.0),
        "cosh": (np.concatenate([rng.uniform(-710, 710, N), 10.0 ** rng.uniform(-20, 1, N)]), mp.cosh, 2.5),
    }
    for name, (x, f, bound) in cases.items():
        worst = _ulps(mathlib(name, x), [f(mp.mpf(float(v))) for v in x])
        assert worst < bound, (name, worst)
    # arguments next to multiples of pi/2: the three-word reduction must not lose the small remainder
    xs = np.array([float(mp.mpf(k) * mp.pi / 2) for k in list(range(1, 400)) + [10 ** 5 + 7, 10 ** 7 + 1, 2 ** 40 + 12345]])
    for name, f in (("sin", mp.sin), ("cos", mp.cos)):
        assert _ulps(mathlib(name, xs), [f(mp.mpf(float(v))) for v in xs]) < 1.0
    x = np.concatenate([10.0 ** rng.uniform(-3, 3, N), 10.0 ** rng.uniform(-300, 300, N), 10.0 ** rng.uniform(-2, 2, N)])
    y = np.concatenate([rng.uniform(-5, 5, N), rng.uniform(-1, 1, N), rng.uniform(-150, 150, N)])
    ref = [mp.power(mp.mpf(float(a)), mp.mpf(float(b))) for a, b in zip(x, y)]
    assert _ulps(mathlib("pow", x, y), ref) < 4.0
    a, b = rng.uniform(-50, 50, N), rng.uniform(-50, 50, N)
    np.testing.assert_allclose(mathlib("logaddexp", a, b), np.logaddexp(a, b), rtol=1e-15)
    np.testing.assert_allclose(mathlib("expit", a), 1 / (1 + np.exp(-a)), rtol=2e-15)
    np.testing.assert_allclose(mathlib("dexpit", a), 1 / ((1 + np.exp(-a)) * (1 + np.exp(a))), rtol=3e-15)


def test_sa_math_special_values(mathlib):
    inf, nan = np.inf, np.nan
    np.testing.assert_array_equal(mathlib("exp", [nan, inf, -inf, 0.0, 709.79, -745.14]), [nan, inf, 0.0, 1.0, inf, 0.0])
    assert mathlib("exp", [-745.13])[0] == 5e-324 and np.isfinite(mathlib("exp", [709.78])[0])
    np.testing.assert_array_equal(mathlib("log", [nan, inf, -1.0, 0.0, 1.0]), [nan, inf, nan, -inf, 0.0])
    assert abs(mathlib("log", [5e-324])[0] - (-744.4400719213812)) < 1e-12
    np.testing.assert_array_equal(mathlib("log1p", [-1.0, -2.0, inf, 1e-300, 0.0]), [-inf, nan, inf, 1e-300, 0.0])
    np.testing.assert_array_equal(mathlib("sin", [0.0, inf, 2.0 ** 51, nan]), [0.0, nan, nan, nan])
    np.testing.assert_array_equal(mathlib("cos", [0.0, -inf, 2.0 ** 51]), [1.0, nan, nan])
    np.testing.assert_array_equal(mathlib("tanh", [0.0, 30.0, -30.0, inf, -inf]), [0.0, 1.0, -1.0, 1.0, -1.0])
    x = [2, -8, -2, -2, 0, 0, inf, inf, 0.5, 2, -1, 1, nan, nan, 4]
    y = [0.5, 1 / 3, 3, 2, 2, -1, 2, -2, inf, -inf, inf, nan, 0, 1, 0.5]
    with np.errstate(all="ignore"):
        want = np.float64(x) ** np.float64(y)
    np.testing.assert_array_equal(mathlib("pow", x, y), want)
    # the five pieces of the degree-4 cardinal B-spline: the reference's polynomial pieces (lambdify.py:74-77)
    t = np.linspace(-0.5, 5.5, 121)
    from sunode_amd.symode.problem import _cardinal_bspline
    np.testing.assert_allclose(mathlib("cardinal_bspline4", t), _cardinal_bspline(4, t), rtol=0, atol=3e-14)


def test_helper_classes_are_the_reference_surface_with_working_derivatives():
    import sympy as sy
    import sunode  # noqa: F401  (installs the alias finder)
    from sunode.symode import lambdify as L
    import sunode_amd.symode.lambdify as mine
    assert L is mine
    x, a, b = sy.symbols("x a b", real=True)
    pts = [dict(zip((x, a, b), v)) for v in ((0.3, 1.7, -0.4), (-2.0, 0.5, 3.0), (4.0, -1.2, 0.1))]

    def same(e1, e2):
        for pt in pts:
            v1, v2 = complex(e1.subs(pt).evalf(30)), complex(e2.subs(pt).evalf(30))
            assert abs(v1 - v2) <= 1e-14 * max(1.0, abs(v2)), (e1, e2, pt)

    # derivative rules == sympy's differentiation of the exp forms
    for f in (L.expit(a * x + b), L.dexpit(a * x * x), L.logaddexp(a * x, b - x), L.expit(L.expit(x) * a)):
        plain = f.rewrite(sy.exp) if not f.has(L.logaddexp) else f.rewrite(sy.log)
        assert not plain.has(L.expit, L.dexpit, L.logaddexp)
        for v in (x, a):
            same(sy.diff(f, v).rewrite(sy.exp), sy.diff(plain, v))
        same(sy.diff(f, x, 2).rewrite(sy.exp), sy.diff(plain, x, 2))
    # the spline: value and derivative rule against the piecewise polynomial
    for d in (1, 2, 3, 4):
        B = L.CardinalBSpline(d, x)
        pw = B.as_sympy_expr()
        fB = sy.lambdify(x, sy.diff(L.CardinalBSpline(d, 2 * x - 1), x).replace(
            L.CardinalBSpline, lambda k, u: L.CardinalBSpline(k, u).as_sympy_expr()))
        fP = sy.lambdify(x, sy.diff(pw.subs(x, 2 * x - 1), x))
        for v in 0.5 + (d + 1) / 2 * (np.arange(23) + 0.37) / 23:        # (2 v - 1 never a knot)
            assert abs(fB(v) - fP(v)) < 1e-12
    # interpolation weights sum to one on [lower, upper] (partition of unity), the reference's affine map
    t = sy.Symbol("t", real=True)
    w = sy.symbols("w0:7", real=True)
    spline = L.interpolate_spline(t, w, 2, 12, 4)
    ones = sy.lambdify(t, spline.subs({wi: 1 for wi in w}).replace(
        L.CardinalBSpline, lambda k, u: L.CardinalBSpline(k, u).as_sympy_expr()))
    for v in np.linspace(2, 12, 41):
        assert abs(ones(v) - 1.0) < 1e-13
    assert L.interpolate_spline(t, w, 2, 12, 4, as_pure=True).has(sy.Piecewise)
    # rewrite rules
    import sympy.codegen.rewriting as rw
    assert rw.optimize(sy.log(sy.exp(a) + sy.exp(b)), [L.logsumexp_2terms_opt]) == L.logaddexp(a, b)
    soft = sy.exp(a) / (sy.exp(a) + sy.exp(b))
    assert L.is_multiple_exp_sum_pow_mult(soft) and not L.is_multiple_exp_sum_pow_mult(sy.exp(a))
    out = L.simplify_multiple_exp_sum(soft)
    assert out.has(L.logaddexp)
    same(out.rewrite(sy.log), soft)
    assert rw.optimize(soft, [L.explog_opt]).has(L.logaddexp)


def test_cse_temporaries_step_aside_for_model_symbols():
    """Parameters named like the temporaries' prefixes (r_, j_, a_, q_, b_, s_ + index): generated C == host lambdify."""
    import sympy as sym
    from oracle.harness import Oracle
    from sunode_amd import SympyProblem

    def rhs(t, y, p):
        g = sym.sin(p.r[1] * y.j[0]) + p.a[2] * y.j[1]
        return {"j": [-p.r[0] * y.j[0] * g + p.q[1], p.b[0] * g * g - p.s[1] * y.j[1] * sym.exp(-p.a[0] * y.j[0])]}
    prob = SympyProblem({"r": (2,), "a": (3,), "q": (2,), "b": (2,), "s": (2,)}, {"j": (2,)}, rhs,
                        [("r",), ("a",), ("s",)])
    orc = Oracle(prob, tag="collide")
    rng = np.random.RandomState(5)
    ud = prob.make_user_data()
    for _ in range(4):
        par = rng.uniform(0.5, 1.5, 11)
        pv = np.zeros((), dtype=prob.params_dtype)
        pv.reshape(1).view(np.float64)[:] = par
        prob.update_params(ud, pv)
        y, lam, t = rng.uniform(0.5, 1.5, 2), rng.randn(2), 0.3
        got = orc.eval(t, y, lam, par[prob.params_subset.subset_index], par[prob.params_subset.remainder_index])
        want = {k: np.zeros_like(np.asarray(got[k], float)) for k in ("rhs", "jac", "adj", "quad", "adjjac")}
        prob.make_rhs()(want["rhs"], t, y, ud)
        prob.make_jac_dense()(want["jac"], t, y, None, ud)
        prob.make_adjoint_rhs()(want["adj"], t, y, lam, ud)
        prob.make_adjoint_quad_rhs()(want["quad"], t, y, lam, ud)
        prob.make_adjoint_jac_dense()(want["adjjac"], t, y, lam, None, ud)
        for k in want:
            np.testing.assert_allclose(got[k], want[k].reshape(np.shape(got[k])), rtol=1e-13, atol=1e-15, err_msg=k)


def test_logistic_switch_callbacks_against_an_independent_derivation():
    """``logistic_switch`` has no reference fixture (the reference cannot differentiate expit of a state): its
    generated callbacks are pinned against sympy's differentiation of the SAME model with every helper rewritten
    into exp / piecewise-polynomial form BEFORE differentiating -- nothing of symode/lambdify.py's fdiff rules or of
    the code generator's printers is on that side."""
    import mpmath as mp
    import sympy as sy
    import sunode_amd.symode.lambdify as L
    prob = make_problem("logistic_switch")
    orc = make_oracle("logistic_switch")

    def plain(e):
        e = sy.sympify(e).replace(L.CardinalBSpline, lambda k, u: L.CardinalBSpline(k, u).as_sympy_expr())
        return e.rewrite(sy.exp)
    ys, ps_s, pr_s = list(prob._sym_statevec), list(prob._sym_deriv_paramsvec), list(prob._sym_fixed_paramsvec)
    f = sy.Matrix([plain(e) for e in prob._sym_dydt])
    J, P = f.jacobian(ys), f.jacobian(ps_s)
    rng = np.random.RandomState(11)
    for _ in range(6):
        y = rng.uniform(0.3, 2.5, 2)
        ps = np.array([np.log(0.9), np.log(4.0), 2.0, 1.5, 1.5]) + 0.2 * rng.randn(5)
        pr, lam = np.array([0.7]), rng.randn(2)
        subs = dict(zip(ys + ps_s + pr_s, [mp.mpf(float(v)) for v in np.concatenate([y, ps, pr])]))
        Jv = np.array(J.subs(subs).evalf(30), dtype=float).reshape(2, 2)
        Pv = np.array(P.subs(subs).evalf(30), dtype=float).reshape(2, 5)
        fv = np.array(f.subs(subs).evalf(30), dtype=float).ravel()
        got = orc.eval(0.0, y, lam, ps, pr)
        np.testing.assert_allclose(got["rhs"], fv, rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(np.asarray(got["jac"]).reshape(2, 2), Jv, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(got["adj"], -lam @ Jv, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(got["quad"], lam @ Pv, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(np.asarray(got["adjjac"]).reshape(2, 2), -Jv.T, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("name", ["forcing", "logistic_switch", "misc"])
def test_oracle_forward_adjoint_matches_truth_on_transcendental_models(name, golden_dir):
    """SURVEY 8(c) bars at rtol = atol = 1e-8: states <= 1e-5, gradients <= 4e-6 relative to DOP853 truth."""
    d = np.load(os.path.join(golden_dir, "truth_%s.npz" % name))
    orc = make_oracle(name)
    cfg = orc.config(rtol=1e-8, atol=1e-8, rtolB=1e-8, atolB=1e-8, rtolQB=1e-8, atolQB=1e-8)
    tv = d["tvals"]
    y, st, _ = orc.solve_forward(cfg, d["y0"], d["ps"], d["pr"], float(d["t0"]), tv, nthreads=4)
    g, lam, stb, _ = orc.solve_backward(cfg, tv[-1], float(d["t0"]), tv, d["grads"], nthreads=4)
    assert (st == 0).all() and (stb == 0).all()
    assert np.max(np.abs(y - d["y_out"]) / np.abs(d["y_out"]).max(axis=(0, 1))) < 1e-5
    assert np.max(np.abs(g - d["grad_params"]) / np.abs(d["grad_params"]).max(axis=1, keepdims=True)) < 4e-6
    assert np.max(np.abs(-lam - d["grad_y0"]) / np.abs(d["grad_y0"]).max(axis=1, keepdims=True)) < 4e-6


def test_generated_header_embeds_the_math_library_only_when_needed():
    from sunode_amd.symode import codegen
    assert "SA_HAVE_MATH" not in make_problem("lv").native_source()
    for name in ("misc", "forcing", "logistic_switch"):
        src = make_problem(name).native_source()
        assert "SA_HAVE_MATH" in src and not codegen.libm_calls(src)
        body = src.split("#endif /* SA_MATH_H */")[1]
        for libm in ("exp(", "log(", "sin(", "cos(", "pow(", "tanh(", "log1p("):     # every call goes to sa_*
            assert not __import__("re").search(r"(?<![\w.])%s" % __import__("re").escape(libm), body), libm
