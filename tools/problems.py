"""Benchmark / test problem definitions in the reference's ``rhs_sympy(t, y, p)`` form.

Config numbering follows BASELINE.json; definitions follow SURVEY.md Appendix D.
Only config 1/2 (Lotka-Volterra) comes from the reference
(/root/reference/README.md:57-91); the others are BASELINE-specified workloads.
The module has no dependency on the engine so that the golden-vector generator
can feed the *same* definitions to the reference's own ``SympyProblem``.
"""
from __future__ import annotations

import numpy as np

SEED = 20260928


# --------------------------------------------------------------------------
# problem definitions: name -> dict(params, states, rhs, derivative_params)
# --------------------------------------------------------------------------
def lotka_volterra(t, y, p):
    return {
        "hares": p.alpha * y.hares - p.beta * y.lynx * y.hares,
        "lynx": p.delta * y.hares * y.lynx - p.gamma * y.lynx,
    }


def robertson(t, y, p):
    return {
        "y1": -p.k1 * y.y1 + p.k2 * y.y2 * y.y3,
        "y2": p.k1 * y.y1 - p.k2 * y.y2 * y.y3 - p.k3 * y.y2 ** 2,
        "y3": p.k3 * y.y2 ** 2,
    }


def robertson5(t, y, p):
    """Robertson with two slow tracer states fed by y1 (five states, four differentiated parameters): a STIFF model at the
    upper end of the one-lane mapping -- its backward steps move over many stored points, i.e. the far path of the index
    search (csrc/bdf_kernels.hip search_left / search_right) at another vector length than config 3's."""
    return {
        "y1": -p.k1 * y.y1 + p.k2 * y.y2 * y.y3,
        "y2": p.k1 * y.y1 - p.k2 * y.y2 * y.y3 - p.k3 * y.y2 ** 2,
        "y3": p.k3 * y.y2 ** 2,
        "w1": p.k1 * y.y1 - p.r * y.w1,
        "w2": p.r * y.w1 * y.y3,
    }


def seir(t, y, p):
    N = y.S + y.E + y.I + y.R
    force = [sum(p.beta[i] * p.C[i, j] * y.I[j] / N[j] for j in range(4)) for i in range(4)]
    return {
        "S": [-force[i] * y.S[i] for i in range(4)],
        "E": [force[i] * y.S[i] - p.rates.sigma * y.E[i] for i in range(4)],
        "I": [p.rates.sigma * y.E[i] - p.rates.gamma * y.I[i] for i in range(4)],
        "R": [p.rates.gamma * y.I[i] for i in range(4)],
    }


def make_network(n):
    def network(t, y, p):
        x = y.x
        tot = sum(x)
        return {"x": [sum(p.K[i, j] * x[j] for j in range(n))
                      - p.scale[0] * x[i] * sum(p.K[j, i] for j in range(n))
                      - p.scale[1] * x[i] * tot / (p.scale[2] + tot) + p.scale[3] for i in range(n)]}
    return network


def notebook_linear(t, y, params):
    """notebooks/from_sympy.ipynb cell 2."""
    return {
        "a": params.c.d * y.a + params.f[20],
        "b": {"c": [3.0, 4.0]},
    }


def misc_functions(t, y, p):
    """Exercises the printer surface ('codegen surface')."""
    import sympy as sym
    return {
        "u": -p.a * y.u ** 2 + sym.exp(-p.b * y.v) * sym.sin(t) + sym.sqrt(y.u) / (1 + y.v ** 3),
        "v": p.a * sym.log(1 + y.u) - y.v ** sym.Rational(3, 2) + sym.tanh(p.c[1] * y.u) - p.c[0] * sym.cos(y.v),
    }


def forcing(t, y, p):
    """The normal PyMC model shape: log-parameterised rates, a smooth switch, a soft threshold and a time-varying
    input through B-spline coefficients -- every helper of the reference's symode/lambdify.py
    (/root/reference/sunode/symode/lambdify.py:275-352) in one right-hand side.  Written so that the REFERENCE can
    derive it too (its expit has no usable derivative rule: the switch depends on time and fixed parameters only),
    which is what makes reference-generated callback fixtures possible (tools/make_golden_callbacks.py)."""
    import sympy as sym
    from sunode.symode.lambdify import expit, interpolate_spline, logaddexp
    r, K = sym.exp(p.log_r), sym.exp(p.log_K)
    u = interpolate_spline(t, list(p.w), 0, 10, 4)
    eaten = p.a * y.x * y.z / (1 + y.z)
    return {
        "x": r * y.x * (1 - y.x / K) * expit(p.k * (t - p.t_mid)) + u - eaten,
        "z": logaddexp(sym.Integer(0), p.s * (y.x - y.z)) - r * y.z / 2,
        "c": eaten,
    }


def logistic_switch(t, y, p):
    """expit of a STATE and of differentiated parameters (d expit = dexpit, d dexpit = dexpit (1 - 2 expit): the rules
    sunode_amd/symode/lambdify.py adds; the reference raises NameError on this model, lambdify.py:301,318), a spline
    of a state (d B_4 = B_3 - shifted B_3) and a general power x^q with a differentiated exponent."""
    import sympy as sym
    from sunode.symode.lambdify import CardinalBSpline, expit
    r, K = sym.exp(p.log_r), sym.exp(p.log_K)
    gate = expit(p.k * (y.x - p.x_half))
    return {
        "x": r * y.x * (1 - y.x / K) - p.m * y.x * gate + CardinalBSpline(4, y.v + 2) / 4,
        "v": gate - r * y.v ** p.q / 3,
    }


def mathfn_a(t, y, p):
    """One deterministic function of csrc/sa_math.h per output (exp / log / log1p / expm1 / pow): the callback-level
    pin of the function library -- device vs oracle bit for bit, both vs the reference's numpy values."""
    import sympy as sym
    from sympy.codegen.cfunctions import expm1, log1p
    x, a = y.x, p.a
    return {"x": [sym.exp(a[0] * x[0] - 3), sym.log(x[1]) + log1p(a[1] * x[1]), x[2] ** a[2],
                  expm1(-a[3] * x[3]) + x[3] ** sym.Rational(-5, 3), 1 / (1 + x[4] ** sym.Rational(7, 2)) + a[4]]}


def mathfn_b(t, y, p):
    """... sin / cos / tan / tanh / sinh / cosh."""
    import sympy as sym
    x, a = y.x, p.a
    return {"x": [sym.sin(10 * a[0] * x[0] + t), sym.cos(x[1] + a[1]), sym.tan(x[2] / 4), sym.tanh(a[3] * x[3] - 1),
                  sym.sinh(x[4]) - sym.cosh(a[4])]}


def huge_pivots(t, y, p):
    """Decay rates of 1e200 on components that are exactly zero: the Newton matrix I - gamma*J has diagonal entries
    around 1e200 (beyond 2^500) while the steps stay of order 1 -- the pivots' reciprocals leave the range in which the
    workgroup LU's lean reciprocal is allowed (csrc/bdf_wave.hip, setup_lu_regs: exponent guard -> general code)."""
    x = y.x
    return {"x": [-p.k[0] * x[0] + 1.0,
                  -p.w[0] * x[1], -p.w[0] * x[2] + p.w[1] * x[1],
                  -p.w[0] * x[3] - p.k[1] * x[4], -p.w[0] * x[4],
                  -x[5] + x[0]]}


def pivoting(t, y, p):
    """Fast rotations far below the tolerances next to slow dynamics: once the step grows, I - gamma*J has
    off-diagonal entries far larger than its diagonal, so the dense LU must exchange rows."""
    x = y.x
    return {"x": [-p.k[0] * x[0] + 1.0,
                  p.w[0] * x[2], -p.w[0] * x[1],
                  p.w[1] * x[4] - p.k[1] * x[3], -p.w[1] * x[3],
                  -x[5] + x[0]]}


PROBLEMS = {
    "lv": dict(
        params={"alpha": (), "beta": (), "gamma": (), "delta": ()},
        states={"hares": (), "lynx": ()},
        rhs=lotka_volterra,
        derivative_params=[("alpha",), ("beta",)],
    ),
    "robertson": dict(
        params={"k1": (), "k2": (), "k3": ()},
        states={"y1": (), "y2": (), "y3": ()},
        rhs=robertson,
        derivative_params=[("k1",), ("k2",), ("k3",)],
    ),
    "robertson5": dict(
        params={"k1": (), "k2": (), "k3": (), "r": ()},
        states={"y1": (), "y2": (), "y3": (), "w1": (), "w2": ()},
        rhs=robertson5,
        derivative_params=[("k1",), ("k2",), ("k3",), ("r",)],
    ),
    "seir": dict(
        params={"beta": (4,), "C": (4, 4), "rates": {"sigma": (), "gamma": (), "mu": (), "nu": ()}},
        states={"S": (4,), "E": (4,), "I": (4,), "R": (4,)},
        rhs=seir,
        derivative_params=[("beta",), ("rates", "sigma"), ("rates", "gamma"), ("rates", "mu"), ("rates", "nu")],
    ),
    "network8": dict(
        params={"K": (8, 8), "scale": (4,)},
        states={"x": (8,)},
        rhs=make_network(8),
        derivative_params=[("scale",)],
    ),
    "notebook": dict(
        params={"c": {"d": (3,)}, "f": (50,)},
        states={"a": (3,), "b": {"c": (2,)}},
        rhs=notebook_linear,
        derivative_params=[("c", "d")],
    ),
    "misc": dict(
        params={"a": (), "b": (), "c": (2,)},
        states={"u": (), "v": ()},
        rhs=misc_functions,
        derivative_params=[("a",), ("c",)],
    ),
    "forcing": dict(
        params={"log_r": (), "log_K": (), "w": (5,), "k": (), "t_mid": (), "a": (), "s": ()},
        states={"x": (), "z": (), "c": ()},
        rhs=forcing,
        derivative_params=[("log_r",), ("log_K",), ("w",), ("a",)],
    ),
    "mathfn_a": dict(params={"a": (5,)}, states={"x": (5,)}, rhs=mathfn_a, derivative_params=[("a",)]),
    "mathfn_b": dict(params={"a": (5,)}, states={"x": (5,)}, rhs=mathfn_b, derivative_params=[("a",)]),
}


def sqrt_decay(t, y, p):
    """x' = -k sqrt(x), z' = -z: x reaches zero at t = 2 sqrt(x0)/k and the right-hand side stops being real:
    a recoverable failure (non-finite rhs) the integrator answers with smaller and smaller steps."""
    import sympy as sym
    return {"x": -p.k * sym.sqrt(y.x), "z": -y.z}


def switched(t, y, p):
    """Discontinuous forcing (Heaviside switches): the error test fails repeatedly at every switch, the order
    drops to 1 and the derivative column is reloaded (cvDoErrorTest's third branch)."""
    import sympy as sym
    return {"x": -p.k * y.x + p.a * sym.Heaviside(t - 1) - 3 * p.a * sym.Heaviside(t - 2),
            "z": y.x - y.z}


def sir_two_groups(t, y, p):
    """Two age groups x (S, I): the smallest model with GROUP structure (the code generator emits its callbacks as lane
    families, symode/codegen.py find_lane_families); four states: runs in the one-lane-per-instance kernel, where the
    family loop is unrolled at compile time."""
    n = [y.S[i] + y.I[i] + p.pop[i] for i in range(2)]
    force = [sum(p.beta[i] * p.C[i, j] * y.I[j] / n[j] for j in range(2)) for i in range(2)]
    return {"S": [-force[i] * y.S[i] + p.gamma * y.I[i] / 4 for i in range(2)],
            "I": [force[i] * y.S[i] - p.gamma * y.I[i] for i in range(2)]}


#: test-only problems without reference-generated golden fixtures
EXTRA_PROBLEMS = {
    "logistic_switch": dict(
        params={"log_r": (), "log_K": (), "k": (), "x_half": (), "q": (), "m": ()},
        states={"x": (), "v": ()},
        rhs=logistic_switch,
        derivative_params=[("log_r",), ("log_K",), ("k",), ("x_half",), ("q",)],
    ),
    "sir2": dict(
        params={"beta": (2,), "C": (2, 2), "gamma": (), "pop": (2,)},
        states={"S": (2,), "I": (2,)},
        rhs=sir_two_groups,
        derivative_params=[("beta",), ("gamma",)],
    ),
    "switched": dict(
        params={"k": (), "a": ()},
        states={"x": (), "z": ()},
        rhs=switched,
        derivative_params=[("k",), ("a",)],
    ),
    "sqrt_decay": dict(
        params={"k": ()},
        states={"x": (), "z": ()},
        rhs=sqrt_decay,
        derivative_params=[("k",)],
    ),
    # dense fixed rate matrix x state: large enough (24 x 24) for the matrix-vector form of the callbacks
    # (symode/problem.py extract_matvec), small enough for every lane-group mapping
    "network24": dict(
        params={"K": (24, 24), "scale": (4,)},
        states={"x": (24,)},
        rhs=make_network(24),
        derivative_params=[("scale",)],
    ),
    "huge_pivots": dict(
        params={"k": (2,), "w": (2,)},
        states={"x": (6,)},
        rhs=huge_pivots,
        derivative_params=[("k",)],
    ),
    "pivoting": dict(
        params={"k": (2,), "w": (2,)},
        states={"x": (6,)},
        rhs=pivoting,
        derivative_params=[("k",)],
    ),
}


def network100():
    return dict(
        params={"K": (100, 100), "scale": (4,)},
        states={"x": (100,)},
        rhs=make_network(100),
        derivative_params=[("scale",)],
    )


# --------------------------------------------------------------------------
# repo-owned counter-based generator: splitmix64 -> Box-Muller
# --------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, idx: np.ndarray) -> np.ndarray:
    """splitmix64 output for counter ``idx`` (vectorised, uint64 wraparound)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + (np.asarray(idx, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def std_normal(seed: int, stream: int, n: int, idx=None) -> np.ndarray:
    """n standard normals; element i depends only on (seed, stream, i).  ``idx``: only these elements (a rank's
    shard of a global batch is generated without materialising the batch)."""
    idx = np.arange(n, dtype=np.uint64) if idx is None else np.asarray(idx, dtype=np.uint64)
    base = np.uint64(stream) << np.uint64(40)
    u1 = (splitmix64(seed, base + np.uint64(2) * idx) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    u2 = (splitmix64(seed, base + np.uint64(2) * idx + np.uint64(1)) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    u1 = np.maximum(u1, 1.0 / 9007199254740992.0)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


# --------------------------------------------------------------------------
# synthetic batches (BASELINE.json configs 2-5)
# --------------------------------------------------------------------------
def lv_batch(B: int, seed: int = SEED, idx=None):
    """Config 2: p = p0*exp(0.25 z), y0 = (1, 0.1)*exp(0.1 z').  Returns full params [B,4]
    in declaration order (alpha, beta, gamma, delta), y0 [B,2], tvals, t0.
    ``idx`` (all batch generators): the global draw indices wanted -- rows idx of the B-draw batch."""
    p0 = np.array([0.1, 0.2, 0.3, 0.4])
    z = np.stack([std_normal(seed, s, B, idx) for s in range(4)], axis=1)
    zy = np.stack([std_normal(seed, 8 + s, B, idx) for s in range(2)], axis=1)
    params = p0 * np.exp(0.25 * z)
    y0 = np.array([1.0, 0.1]) * np.exp(0.1 * zy)
    return dict(params=params, y0=y0, tvals=np.linspace(0, 10), t0=0.0,
                rtol=1e-8, atol=1e-8)


def robertson_batch(B: int, seed: int = SEED, idx=None):
    """Config 3: k = k0*exp(0.1 z), y0 = (1, 0, 0), T = 4e4, rtol 1e-8 / atol 1e-10."""
    k0 = np.array([0.04, 1e4, 3e7])
    z = np.stack([std_normal(seed, 16 + s, B, idx) for s in range(3)], axis=1)
    params = k0 * np.exp(0.1 * z)
    y0 = np.tile(np.array([1.0, 0.0, 0.0]), (len(z), 1))
    tvals = np.array([0.0] + [0.4 * 10.0 ** k for k in range(6)])
    return dict(params=params, y0=y0, tvals=tvals, t0=0.0, rtol=1e-8, atol=1e-10)


def robertson5_batch(B: int, seed: int = SEED, idx=None):
    """robertson5: k = k0*exp(0.1 z), r = 0.05*exp(0.2 z'), y0 = (1, 0, 0, 0, 0), T = 4e3, rtol 1e-8 / atol 1e-10."""
    k0 = np.array([0.04, 1e4, 3e7])
    z = np.stack([std_normal(seed, 80 + s, B, idx) for s in range(4)], axis=1)
    params = np.concatenate([k0 * np.exp(0.1 * z[:, :3]), 0.05 * np.exp(0.2 * z[:, 3:])], axis=1)
    y0 = np.tile(np.array([1.0, 0.0, 0.0, 0.0, 0.0]), (len(z), 1))
    tvals = np.array([0.0] + [0.4 * 10.0 ** k for k in range(5)])
    return dict(params=params, y0=y0, tvals=tvals, t0=0.0, rtol=1e-8, atol=1e-10)


def seir_batch(B: int, seed: int = SEED, idx=None):
    """Config 4: 4 groups x (S,E,I,R); beta + 4 rates differentiated, C shared."""
    beta0 = np.array([0.30, 0.25, 0.35, 0.20])
    rates0 = np.array([0.2, 0.1, 0.01, 0.02])          # sigma, gamma, mu, nu
    C = np.array([[1.0, 0.3, 0.2, 0.1],
                  [0.3, 1.0, 0.3, 0.2],
                  [0.2, 0.3, 1.0, 0.3],
                  [0.1, 0.2, 0.3, 1.0]])
    z = np.stack([std_normal(seed, 32 + s, B, idx) for s in range(8)], axis=1)
    sub = np.concatenate([beta0, rates0]) * np.exp(0.1 * z)          # [B, 8] subset order
    pop = np.array([1000.0, 800.0, 1200.0, 600.0])
    I0 = np.array([1.0, 0.0, 2.0, 0.0])
    y0 = np.concatenate([pop - I0, np.zeros(4), I0, np.zeros(4)])
    return dict(ps=sub, pr=C.ravel(), y0=np.tile(y0, (len(z), 1)), tvals=np.linspace(0, 100, 51), t0=0.0,
                rtol=1e-8, atol=1e-8)


def _cotangents(B, n_t, n, idx=None):
    """Per-instance cotangents dL/dy_out (non-degenerate, different for every draw)."""
    k = np.arange(n_t)[None, :, None]
    i = np.arange(n)[None, None, :]
    b = np.arange(B)[:, None, None] if idx is None else np.asarray(idx)[:, None, None]
    return 1.0 + 0.5 * np.cos(1.3 * k + 0.7 * i + 0.21 * b)


def forcing_batch(B: int, seed: int = SEED, idx=None):
    """Draws for ``forcing``: subset order (log_r, log_K, w[5], a), remainder (k, t_mid, s)."""
    z = np.stack([std_normal(seed, 1200 + k, B, idx) for k in range(8)], axis=1)
    base = np.array([np.log(0.8), np.log(5.0), 0.2, 0.5, 1.0, 0.3, 0.6, 0.4])
    ps = base + 0.2 * z
    ps[:, 2:7] = base[2:7] * np.exp(0.2 * z[:, 2:7])
    ps[:, 7] = base[7] * np.exp(0.2 * z[:, 7])
    zy = np.stack([std_normal(seed, 1220 + s, B, idx) for s in range(2)], axis=1)
    y0 = np.concatenate([np.array([0.5, 0.2]) * np.exp(0.1 * zy), np.zeros((len(z), 1))], axis=1)
    tvals = np.linspace(0, 10, 11)
    return dict(ps=ps, pr=np.array([2.0, 3.0, 1.5]), y0=y0, tvals=tvals, t0=0.0,
                grads=_cotangents(len(z), len(tvals), 3, idx), rtol=1e-8, atol=1e-8)


def logistic_switch_batch(B: int, seed: int = SEED, idx=None):
    """Draws for ``logistic_switch``: subset (log_r, log_K, k, x_half, q), remainder (m)."""
    z = np.stack([std_normal(seed, 1240 + k, B, idx) for k in range(5)], axis=1)
    base = np.array([np.log(0.9), np.log(4.0), 2.0, 1.5, 1.5])
    ps = base + 0.1 * z
    zy = np.stack([std_normal(seed, 1250 + s, B, idx) for s in range(2)], axis=1)
    y0 = np.array([0.4, 0.3]) * np.exp(0.1 * zy)
    tvals = np.linspace(0, 8, 9)
    return dict(ps=ps, pr=np.array([0.7]), y0=y0, tvals=tvals, t0=0.0,
                grads=_cotangents(len(z), len(tvals), 2, idx), rtol=1e-8, atol=1e-8)


def misc_batch(B: int, seed: int = SEED, idx=None):
    """Draws for ``misc`` (exp / sin / sqrt / log / x^(3/2) / tanh / cos): subset (a, c[2]), remainder (b)."""
    z = np.stack([std_normal(seed, 1260 + k, B, idx) for k in range(3)], axis=1)
    ps = np.array([0.5, 0.2, 1.0]) * np.exp(0.1 * z)
    zy = np.stack([std_normal(seed, 1270 + s, B, idx) for s in range(2)], axis=1)
    y0 = np.array([1.0, 0.5]) * np.exp(0.1 * zy)
    tvals = np.linspace(0, 12, 13)
    return dict(ps=ps, pr=np.array([2.0]), y0=y0, tvals=tvals, t0=0.0,
                grads=_cotangents(len(z), len(tvals), 2, idx), rtol=1e-8, atol=1e-8)


def network_batch(B: int, n: int = 100, seed: int = SEED, idx=None):
    """Config 5: dense mass-action network, K (n x n) shared fixed, scale(4) differentiated per draw."""
    u = np.abs(std_normal(seed, 64, n * n)).reshape(n, n)
    K = u / n
    scale0 = np.array([1.0, 0.5, 10.0, 0.1])
    z = np.stack([std_normal(seed, 65 + s, B, idx) for s in range(4)], axis=1)
    ps = scale0 * np.exp(0.1 * z)
    zx = std_normal(seed, 70, n)
    y0 = np.tile(np.exp(0.3 * zx), (len(z), 1))
    return dict(ps=ps, pr=K.ravel(), y0=y0, tvals=np.linspace(0, 10, 11), t0=0.0, rtol=1e-8, atol=1e-8)


# --------------------------------------------------------------------------
# generated model family for the shape sweep (tests/test_shape_sweep.py): any (n states, p differentiated parameters)
# --------------------------------------------------------------------------
def make_random_network(n: int, p: int, band: int = 0):
    """Mass-action network with n states, a shared fixed rate matrix K (n x n) and p differentiated scales s:

        x_i' = sum_j K_ij x_j  -  A_i x_i sum_j K_ji  -  B_i x_i T / (C + T)  +  D_i,      T = sum_j x_j

    with A_i = sum of the s_k with k = i (mod n) (1 if there is none), B_i = s_{(i+1) mod p} / 2, C = 1 + s_{2 mod p},
    D_i = s_{(3i+1) mod p} / 10 -- every parameter enters, with p > n several per equation, with p < n each one in
    several equations.  ``band`` > 0: only the entries |i - j| <= band of K are used (sparse Jacobian, no dense
    matrix-vector block for the code generator to find)."""
    def rhs(t, y, par):
        import sympy as sym
        x = y.x
        s = par.s
        tot = sum(x)

        def used(i, j):
            return band <= 0 or abs(i - j) <= band
        out = []
        for i in range(n):
            mine = [s[k] for k in range(p) if k % n == i]
            a_i = sum(mine) if mine else sym.Integer(1)
            inflow = sum(par.K[i, j] * x[j] for j in range(n) if used(i, j))
            colsum = sum(par.K[j, i] for j in range(n) if used(j, i))
            out.append(inflow - a_i * x[i] * colsum - s[(i + 1) % p] / 2 * x[i] * tot / (1 + s[2 % p] + tot)
                       + s[(3 * i + 1) % p] / 10)
        return {"x": out}
    return rhs


def make_chain(n: int):
    """Linear decay chain x_0 -> x_1 -> ... with a source: n states, a bidiagonal Jacobian, two differentiated rates.
    Cheap to derive at any size -- the memory-resident mapping's sizes (n > 128) without minutes of sympy."""
    def rhs(t, y, p):
        x = y.x
        return {"x": [-p.k[0] * x[0] + p.k[1]]
                + [p.k[0] * x[i - 1] - (p.k[0] + p.k[1] / (i + 1)) * x[i] for i in range(1, n)]}
    return rhs


def chain(n: int):
    return dict(params={"k": (2,)}, states={"x": (n,)}, rhs=make_chain(n), derivative_params=[("k",)])


def chain_batch(B: int, n: int, seed: int = SEED, idx=None):
    z = np.stack([std_normal(seed, 1300 + s, B, idx) for s in range(2)], axis=1)
    ps = np.array([1.5, 0.8]) * np.exp(0.2 * z)
    y0 = np.zeros((len(z), n))
    y0[:, 0] = 1.0
    tvals = np.array([0.0, 0.5, 1.0, 2.0])
    return dict(ps=ps, pr=np.zeros(0), y0=y0, tvals=tvals, t0=0.0,
                grads=_cotangents(len(z), len(tvals), n, idx), rtol=1e-6, atol=1e-8)


def random_network(n: int, p: int, band: int = 0):
    """Problem specification (the dict ``SympyProblem`` is built from) of the (n, p) member of the family."""
    return dict(params={"K": (n, n), "s": (p,)}, states={"x": (n,)}, rhs=make_random_network(n, p, band),
                derivative_params=[("s",)])


def lv12(t, y, p):
    """Lotka-Volterra with saturating terms: 2 states, 12 differentiated parameters (few states, many parameters:
    the quadrature history, not the state, decides the mapping)."""
    h, l = y.hares, y.lynx
    a, b, c = p.a, p.b, p.c
    return {
        "hares": a[0] * h - b[0] * h * l + a[1] * h / (1 + c[0] * h) - a[2] * h ** 2 / 10 + c[2] / 10,
        "lynx": b[1] * h * l - a[3] * l + b[2] * l / (1 + c[1] * l) - b[3] * l ** 2 / 10 + c[3] / 10,
    }


LV12 = dict(params={"a": (4,), "b": (4,), "c": (4,)}, states={"hares": (), "lynx": ()}, rhs=lv12,
            derivative_params=[("a",), ("b",), ("c",)])


def random_network_batch(B: int, n: int, p: int, seed: int = SEED, idx=None):
    """Draws for ``random_network(n, p)``: K shared (|N(0,1)| / n), s = s0 exp(0.2 z) per draw, y0 = exp(0.3 z') per
    draw, per-instance cotangents; 6 output times on [0, 8]."""
    stream = 1000 + 131 * n + p
    K = np.abs(std_normal(seed, stream, n * n)).reshape(n, n) / n
    s0 = 0.6 + 0.1 * (np.arange(p) % 7)
    z = np.stack([std_normal(seed, stream + 1 + (k % 50), B, idx) for k in range(p)], axis=1) \
        * (1.0 + 0.05 * (np.arange(p) // 50))
    ps = s0 * np.exp(0.2 * z)
    zx = np.stack([std_normal(seed, stream + 60 + (i % 40), B, idx) for i in range(n)], axis=1)
    y0 = np.exp(0.3 * zx * (1.0 + 0.03 * (np.arange(n) // 40)))
    tvals = np.array([0.0, 0.5, 1.5, 3.0, 5.0, 8.0])
    k = np.arange(len(tvals))[None, :, None]
    i = np.arange(n)[None, None, :]
    b = np.arange(len(y0))[:, None, None] if idx is None else np.asarray(idx)[:, None, None]
    grads = 1.0 + 0.5 * np.cos(1.7 * k + 0.9 * i + 0.37 * b)
    return dict(ps=ps, pr=K.ravel(), y0=y0, tvals=tvals, t0=0.0, grads=grads, rtol=1e-8, atol=1e-8)


def lv12_batch(B: int, seed: int = SEED, idx=None):
    z = np.stack([std_normal(seed, 900 + k, B, idx) for k in range(12)], axis=1)
    base = np.array([0.1, 0.05, 0.3, 0.3, 0.2, 0.4, 0.05, 0.2, 0.5, 0.5, 0.1, 0.1])
    ps = base * np.exp(0.2 * z)
    zy = np.stack([std_normal(seed, 920 + s, B, idx) for s in range(2)], axis=1)
    y0 = np.array([1.0, 0.1]) * np.exp(0.1 * zy)
    tvals = np.linspace(0, 10, 11)
    k = np.arange(len(tvals))[None, :, None]
    i = np.arange(2)[None, None, :]
    b = np.arange(len(y0))[:, None, None] if idx is None else np.asarray(idx)[:, None, None]
    grads = 1.0 + 0.5 * np.cos(1.3 * k + 0.7 * i + 0.21 * b)
    return dict(ps=ps, pr=np.zeros(0), y0=y0, tvals=tvals, t0=0.0, grads=grads, rtol=1e-8, atol=1e-8)
