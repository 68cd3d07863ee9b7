"""Parity over the model-shape space: the DEFAULT build of every (n, p) band and boundary vs the CPU oracle.

The product is a code generator: every user model is a new code object whose kernel family follows from (n, p)
(sunode_amd/_native.py ``kernel_variant``).  The BASELINE problems cover n in {2, 3, 16, 100}; this sweep runs a
generated model family (tools/problems.py ``random_network`` -- the reference takes any sympy system,
/root/reference/sunode/symode/problem.py:25-33) through the mapping the engine selects BY ITSELF (no SA_FORCE_GROUP)
on both sides of every selection boundary.  Bar: statuses, every counter and every fp64 output ``array_equal``.
"""
import os

import numpy as np
import pytest

from tests.helpers import make_oracle, make_problem
from tools.sweep_cases import ADJOINT_CASES, SENS_CASES, batch_of

CMP = [0, 1, 2, 3, 4, 5, 6, 7, 8]
CMP_B = [0, 1, 2, 3, 4, 5, 6, 9, 10, 12]


def test_sweep_covers_every_mapping_family():
    """CPU: the shapes of the sweep reach every kernel family and both sides of every boundary of the selection."""
    from sunode_amd import _native
    seen = set()
    for name, _ in ADJOINT_CASES:
        assert not os.environ.get("SA_FORCE_GROUP")
        prob = make_problem(name) if name in ("lv12", "rn5_8", "rn6_1") else None
        import re
        m = re.fullmatch(r"rnb?(\d+)_(\d+)", name)
        n, p = (2, 12) if name == "lv12" else (int(m.group(1)), int(m.group(2)))
        src = "#define SA_N_STATES %d\n#define SA_N_SUB %d\n" % (n, p)
        fam = _native.kernel_variant(src)
        if prob is not None:
            assert fam == _native.kernel_variant(prob.native_source())
        lean = fam[0] == "bdf_wave.hip" and fam[1] <= 8 and n * ((n + fam[1] - 1) // fam[1]) <= 64
        seen.add((fam[0], fam[1], lean))
    assert {("bdf_kernels.hip", 1, False), ("bdf_wave.hip", 4, True), ("bdf_wave.hip", 8, True),
            ("bdf_wave.hip", 16, False), ("bdf_wave.hip", 32, False), ("bdf_wave.hip", 64, False),
            ("bdf_mem.hip", 1, False)} <= seen


@pytest.mark.gpu
@pytest.mark.parametrize("name,B", ADJOINT_CASES, ids=[c[0] for c in ADJOINT_CASES])
def test_default_mapping_adjoint_equals_oracle(name, B):
    from sunode_amd.solver import AdjointSolver
    assert not os.environ.get("SA_FORCE_GROUP")
    prob = make_problem(name)
    d = batch_of(name, B)
    tol = dict(abstol=d["atol"], reltol=d["rtol"], backward_abstol=d["atol"], backward_reltol=d["rtol"],
               quad_abstol=d["atol"], quad_reltol=d["rtol"])
    sol = AdjointSolver(prob, **tol)
    tv = d["tvals"]
    y, st, stats = sol.solve_forward_batch(d["t0"], tv, d["y0"], d["ps"], d["pr"])
    g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], d["t0"], tv, d["grads"])
    orc = make_oracle(name)
    cfg = orc.config(rtol=d["rtol"], atol=d["atol"], rtolB=d["rtol"], atolB=d["atol"], rtolQB=d["rtol"],
                     atolQB=d["atol"])
    yo, so, sto = orc.solve_forward(cfg, d["y0"], d["ps"], d["pr"], d["t0"], tv, nthreads=8)
    go, lo, sbo, stbo = orc.solve_backward(cfg, tv[-1], d["t0"], tv, d["grads"], nthreads=8)
    assert (so == 0).all() and (sbo == 0).all()
    np.testing.assert_array_equal(st, so)
    np.testing.assert_array_equal(stats[:, CMP], sto[:, CMP])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(stb, sbo)
    np.testing.assert_array_equal(statsb[:, CMP_B], stbo[:, CMP_B])
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)
    assert np.isfinite(g).all() and np.abs(g).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["simultaneous", "staggered"])
@pytest.mark.parametrize("name,B", SENS_CASES, ids=[c[0] for c in SENS_CASES])
def test_default_mapping_sensitivities_equal_oracle(name, B, mode):
    from sunode_amd.solver import Solver
    assert not os.environ.get("SA_FORCE_GROUP")
    prob = make_problem(name)
    d = batch_of(name, B)
    sol = Solver(prob, abstol=d["atol"], reltol=d["rtol"], sens_mode=mode)
    n, p = prob.n_states, prob.n_params
    sens0 = np.zeros((p, n))
    y, s, st, stats = sol.solve_sens_batch(d["t0"], d["tvals"], d["y0"], d["ps"], d["pr"], sens0)
    orc = make_oracle(name)
    cfg = orc.config(rtol=d["rtol"], atol=d["atol"])
    yo, so_, sto_, statso = orc.solve_sens(cfg, d["y0"], d["ps"], d["pr"], sens0, d["t0"], d["tvals"], mode=mode,
                                           nthreads=8)
    assert (sto_ == 0).all()
    np.testing.assert_array_equal(st, sto_)
    np.testing.assert_array_equal(stats[:, CMP[:8]], statso[:, CMP[:8]])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(s, so_)
    assert np.abs(s).max() > 0
