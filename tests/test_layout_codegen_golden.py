"""Layout + generated callbacks vs vectors produced by the REFERENCE's symbolic half
(tests/golden/{layout,callbacks}.json, generator: tools/make_golden_callbacks.py)."""
import json
import os

import numpy as np
import pytest

from tests.helpers import make_oracle, make_problem
from tools.problems import PROBLEMS

NAMES = list(PROBLEMS)


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "layout.json")) as fh:
        layout = json.load(fh)
    with open(os.path.join(golden_dir, "callbacks.json")) as fh:
        callbacks = json.load(fh)
    return layout, callbacks


@pytest.mark.parametrize("name", NAMES)
def test_layout_matches_reference(name, golden):
    ref = golden[0][name]
    prob = make_problem(name)
    ps = prob.params_subset
    assert prob.n_states == ref["n_states"]
    assert prob.n_params == ref["n_params"]
    assert ps.n_items == ref["n_items"]
    assert [list(p) for p in prob.state_subset.paths] == ref["state_paths"]
    assert [list(p) for p in ps.paths] == ref["param_paths"]
    assert [list(p) for p in ps.subset_paths] == ref["subset_paths"]
    assert [list(p) for p in ps.remainder.subset_paths] == ref["remainder_subset_paths"]
    assert {".".join(q): [s.start, s.stop] for q, s in ps.flat_slices.items()} == ref["param_slices"]
    assert {".".join(q): list(s) for q, s in ps.flat_shapes.items()} == ref["param_shapes"]
    assert {".".join(q): [s.start, s.stop] for q, s in prob.state_subset.flat_slices.items()} == ref["state_slices"]
    assert prob.params_dtype.itemsize == ref["params_itemsize"]
    assert prob.state_dtype.itemsize == ref["state_itemsize"]
    assert ps.subset_dtype.itemsize == ref["subset_itemsize"]
    view, rview = ps.subset_view_dtype, ps.remainder.subset_view_dtype
    assert [int(view.fields[n][1]) for n in (view.names or ())] == ref["subset_view_offsets"]
    assert [int(rview.fields[n][1]) for n in (rview.names or ())] == ref["remainder_view_offsets"]
    assert prob.user_data_dtype.itemsize == ref["user_data_itemsize"]
    # index maps are consistent with the slices
    sub = np.concatenate([np.arange(*ref["param_slices"][".".join(p)]) for p in ref["subset_paths"]]) \
        if ref["subset_paths"] else np.zeros(0, int)
    assert ps.subset_index.tolist() == sub.tolist()
    assert sorted(ps.subset_index.tolist() + ps.remainder_index.tolist()) == list(range(ref["n_items"]))


@pytest.mark.parametrize("name", NAMES)
def test_generated_c_callbacks_match_reference(name, golden):
    """Host-compiled generated C (same text the HIP kernels include) vs the reference's lambdify output."""
    prob = make_problem(name)
    orc = make_oracle(name)
    ps_idx, pr_idx = prob.params_subset.subset_index, prob.params_subset.remainder_index
    for pt in golden[1][name]:
        par = np.array(pt["params"])
        got = orc.eval(pt["t"], pt["y"], pt["lam"], par[ps_idx], par[pr_idx])
        for key in ("rhs", "jac", "adj", "quad", "adjjac"):
            want = np.array(pt[key], dtype=float)
            # expression order differs (CSE / expanded powers): allow rounding relative to the
            # largest entry of the same output (cancellation), nothing more
            scale = float(np.max(np.abs(want))) if want.size else 0.0
            np.testing.assert_allclose(got[key], want.reshape(got[key].shape), rtol=1e-13,
                                       atol=4e-15 * scale, err_msg="%s/%s" % (name, key))
        assert got["codes"].tolist() == pt["codes"]


@pytest.mark.parametrize("name", ["lv", "seir", "misc"])
def test_host_callables_match_reference(name, golden):
    """make_rhs()/make_jac_dense()/... host conveniences keep the reference's call signatures."""
    prob = make_problem(name)
    n, p = prob.n_states, prob.n_params
    for pt in golden[1][name][:3]:
        ud = prob.make_user_data()
        pv = np.zeros((), dtype=prob.params_dtype)
        pv.reshape(1).view(np.float64)[:] = pt["params"]
        prob.update_params(ud, pv)
        y = np.zeros((), dtype=prob.state_dtype)
        y.reshape(1).view(np.float64)[:] = pt["y"]
        lam = np.array(pt["lam"])
        out = np.zeros(n)
        assert prob.make_rhs()(out, pt["t"], y, ud) == pt["codes"][0]
        np.testing.assert_allclose(out, pt["rhs"], rtol=1e-13)
        J = np.zeros((n, n))
        prob.make_jac_dense()(J, pt["t"], y, None, ud)
        np.testing.assert_allclose(J, pt["jac"], rtol=1e-13)
        prob.make_adjoint_rhs()(out, pt["t"], y, lam, ud)
        np.testing.assert_allclose(out, pt["adj"], rtol=1e-12, atol=1e-14)
        q = np.zeros(p)
        prob.make_adjoint_quad_rhs()(q, pt["t"], y, lam, ud)
        np.testing.assert_allclose(q, pt["quad"], rtol=1e-12, atol=1e-14)
        prob.make_adjoint_jac_dense()(J, pt["t"], y, lam, None, ud)
        np.testing.assert_allclose(J, pt["adjjac"], rtol=1e-13)


def test_nonfinite_is_recoverable_error():
    """Reference symode/problem.py:266-269: non-finite output -> return code 1."""
    orc = make_oracle("misc")
    got = orc.eval(0.1, [-1.0, 0.5], [1.0, 1.0], [1.0, 1.0, 1.0], [1.0])   # sqrt(-1), log(0)
    assert got["codes"][0] == 1


def test_matvec_form_of_dense_linear_blocks():
    """A dense (fixed parameter) x (state | adjoint state) block is emitted as ONE matrix-vector product
    (SA_MATVEC) + a shared dot product + short residues.  Pure re-grouping: the generated C (compiled into the
    oracle) must equal sympy's evaluation of the ORIGINAL expressions to rounding."""
    from oracle.harness import Oracle
    from tests.helpers import make_problem
    prob = make_problem("network24")
    src = prob.native_source()
    assert set(prob._matvec) == {"f", "a"} and prob._matvec["a"]["n_in"] == 24
    assert "SA_MATVEC(a, 24, 24," in src and "SA_MATVEC(f, 24, 24," in src and "SA_OWNS(23)" in src
    assert {k: (v["axis"], len(v["exceptions"])) for k, v in prob._matfill.items()} == {"j": (0, 24), "b": (1, 24)}
    assert "SA_MATFILL(j, 24," in src and "SA_MATFILL(b, 24," in src
    assert prob.n_remainder_native == prob.n_remainder + len(prob._hoisted) + len(prob._packed) + 4 * 24 * 24
    orc = Oracle(prob, "network24")
    rng = np.random.RandomState(3)
    n = 24
    for _ in range(4):
        K = np.abs(rng.randn(n, n)) / n
        sc = np.array([1.0, 0.5, 10.0, 0.1]) * np.exp(0.1 * rng.randn(4))
        y = np.exp(0.3 * rng.randn(n))
        lam = rng.randn(n)
        pr = prob.extend_remainder(K.ravel())
        # block (j, i) of the j-major copies: K[i, j] for the right-hand side, -K[j, i] for the adjoint
        off_f, off_a = prob._matvec["f"]["offset"], prob._matvec["a"]["offset"]
        np.testing.assert_array_equal(pr[off_f:off_f + n * n].reshape(n, n), K.T)
        np.testing.assert_array_equal(pr[off_a:off_a + n * n].reshape(n, n), -K)
        got = orc.eval(0.3, y, lam, sc, pr)
        ud = prob.make_user_data()
        prob.update_params(ud, np.concatenate([K.ravel(), sc]).view(prob.params_dtype)[0])
        ys = y.view(prob.state_dtype)[0]
        want = np.zeros(n); prob.make_rhs()(want, 0.3, ys, ud)
        np.testing.assert_allclose(got["rhs"], want, rtol=1e-13, atol=1e-15)
        want = np.zeros(n); prob.make_adjoint_rhs()(want, 0.3, ys, lam, ud)
        np.testing.assert_allclose(got["adj"], want, rtol=1e-12, atol=1e-14)
        want = np.zeros(4); prob.make_adjoint_quad_rhs()(want, 0.3, ys, lam, ud)
        np.testing.assert_allclose(got["quad"], want, rtol=1e-12, atol=1e-14)
        # matrix callbacks: constant block + line vector + diagonal exceptions (SA_MATFILL)
        want = np.zeros((n, n)); prob.make_jac_dense()(want, 0.3, ys, None, ud)
        np.testing.assert_allclose(got["jac"], want, rtol=1e-13, atol=1e-15)
        want = np.zeros((n, n)); prob.make_adjoint_jac_dense()(want, 0.3, ys, lam, None, ud)
        np.testing.assert_allclose(got["adjjac"], want, rtol=1e-13, atol=1e-15)
        assert got["codes"].tolist() == [0, 0, 0, 0, 0]
    # a problem without such a block keeps the plain form
    assert not make_problem("seir")._matvec and "SA_MATVEC(" not in make_problem("seir").native_source().split("SA_TEMPLATE SA_FN int sa_rhs")[1]


def test_matvec_regrouping_of_an_unexpanded_rhs_is_exact():
    """the same coefficient atom twice for one (output, v_j) pair -- y_j*(g + b) + g*y_j on every row of a
    dense 16 x 16 block, nothing expanded -- used to lose one g*y_j per output (both occurrences removed, the shared
    term added once).  Substituting the matrix-vector symbols back must give the original expressions exactly."""
    import sympy as sym
    from sunode_amd.symode.problem import extract_matvec
    n = 16
    y = sym.symbols("y0:%d" % n, positive=True)
    K = [[sym.Symbol("K_%d_%d" % (i, j), real=True) for j in range(n)] for i in range(n)]
    g, b = sym.symbols("g b", real=True)
    fixed = {K[i][j] for i in range(n) for j in range(n)}
    exprs = [sum(K[i][j] * y[j] for j in range(n)) + y[0] * (g + b) + g * y[0] + (i + 1) * b * y[3] - y[i] ** 2
             for i in range(n)]
    out = extract_matvec(exprs, list(y), fixed, "t")
    assert out is not None
    new, mv, entries = out
    back = {mv[i]: sum(entries[j][i][0] * entries[j][i][1] * y[j] for j in range(n) if entries[j][i] is not None)
            for i in range(n)}
    for i in range(n):
        assert sym.expand(new[i].subs(back) - exprs[i]) == 0, i
    # the shared part really was pulled out once (2*g + b multiplies y0 in every output)
    assert all(sym.expand(new[i]).coeff(y[0]).subs({mv[i]: 0}) == 2 * g + b for i in range(n))


@pytest.mark.parametrize("name", ["network24", "network100"])
def test_structured_network_callbacks_match_reference(name, golden_dir):
    """the callbacks this build emits in STRUCTURED form (SA_MATVEC / SA_MATFILL / SA_SUM / SA_ROLLED) against
    values produced by the REFERENCE's own lambdify pipeline (tests/golden/callbacks_network.json), not a sympy
    re-evaluation of our expressions: vectors entry by entry, the n x n Jacobians through two dense projections, the
    diagonal and a strided sample (n = 100 would be megabytes otherwise).  Generated C compiled into the oracle."""
    from tests.helpers import check_matrix_summary, network_golden_points
    prob = make_problem(name)
    orc = make_oracle(name)
    n, pts = network_golden_points(golden_dir, name)
    for pt in pts:
        got = orc.eval(pt["t"], pt["y"], pt["lam"], np.array(pt["scale"]), pt["K"])
        for key in ("rhs", "adj", "quad"):
            ref = np.array(pt[key])
            np.testing.assert_allclose(got[key], ref, rtol=1e-12, atol=64 * 2.3e-16 * np.abs(ref).max(), err_msg=key)
        check_matrix_summary(got["jac"], pt["jac"])
        check_matrix_summary(got["adjjac"], pt["adjjac"])
        assert got["codes"].tolist() == pt["codes"]


def test_group_structure_becomes_lane_families():
    """symode/codegen.py find_lane_families: a model written as loops over M groups (SEIR: 4 compartments x 4 age groups)
    is equivariant under a relabelling of the groups, so its callbacks are emitted ONCE for group 0 inside SA_FAM_BEGIN(M)
    -- the lean lane groups then evaluate one member per lane (round-3 / round-4 review item).  The generated C (plain
    loop over the members) must give the values of the ORIGINAL sympy expressions; models without the symmetry keep
    the ordinary form."""
    import sympy as sym
    from tests.helpers import make_oracle, make_problem
    seir = make_problem("seir")
    src = seir.native_source()
    body = {name: src[src.index("int %s(" % name):].split("\n}\n")[0] for name in ("sa_rhs", "sa_adj_rhs", "sa_quad_rhs", "sa_jac")}
    assert body["sa_rhs"].count("SA_FAM_BEGIN(4)") == 1 and body["sa_rhs"].count("SA_FAM_STORE(") == 4
    assert body["sa_adj_rhs"].count("SA_FAM_BEGIN(4)") == 1 and body["sa_adj_rhs"].count("SA_FAM_STORE(") == 4
    assert body["sa_quad_rhs"].count("SA_FAM_STORE(") == 1          # beta(4); the four rates are ordinary statements
    assert "SA_FAM_BEGIN" not in body["sa_jac"]
    for name in ("lv", "robertson", "misc"):
        assert "SA_FAM_BEGIN(" not in make_problem(name).native_source().split("#endif")[-1], name
    assert make_problem("network8").native_source().count("SA_FAM_BEGIN(8)") >= 2        # interchangeable species: a family too
    assert make_problem("notebook").native_source().count("SA_FAM_BEGIN(3)") >= 1        # an element-wise vector equation as well
    sir2 = make_problem("sir2")
    assert sir2.native_source().count("SA_FAM_BEGIN(2)") == 3
    # values: the oracle's build of the generated C against sympy's own evaluation of the unrolled expressions
    rng = np.random.RandomState(5)
    for prob, name in ((seir, "seir"), (sir2, "sir2")):
        orc = make_oracle(name)
        n, p = prob.n_states, prob.n_params
        for _ in range(4):
            y, lam = rng.uniform(0.5, 3.0, n), rng.randn(n)
            ps, pr = rng.uniform(0.1, 1.0, p), rng.uniform(0.1, 1.0, prob.n_remainder)
            got = orc.eval(0.3, y, lam, ps, pr)
            sub = dict(zip(prob._sym_statevec, y)); sub.update(zip(prob._sym_lamda, lam))
            sub.update(zip(prob._sym_deriv_paramsvec, ps)); sub.update(zip(prob._sym_fixed_paramsvec, pr))
            for key, exprs in (("rhs", prob._sym_dydt), ("adj", prob._sym_dlamdadt), ("quad", prob._sym_quad_rhs)):
                want = np.array([float(sym.sympify(e).xreplace(sub)) for e in exprs])
                np.testing.assert_allclose(got[key], want, rtol=1e-13, atol=1e-14 * np.abs(want).max())
    # a model whose groups are NOT interchangeable (one group has a term of its own) keeps the ordinary form
    from sunode_amd import SympyProblem

    def lopsided(t, y, p):
        out = sir2._rhs_sympy_func(t, y, p)
        out["I"][1] = out["I"][1] - p.gamma * y.I[1] ** 2
        return out
    spec = dict(params={"beta": (2,), "C": (2, 2), "gamma": (), "pop": (2,)}, states={"S": (2,), "I": (2,)})
    odd = SympyProblem(spec["params"], spec["states"], lopsided, [("beta",), ("gamma",)])
    text = odd.native_source()
    assert text[text.index("int sa_rhs("):text.index("int sa_jac(")].count("SA_FAM_STORE(") == 1    # S still is a family, I is not
