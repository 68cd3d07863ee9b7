"""Full-size and arena tests on the MI355X (pytest -m gpu).

* every BASELINE configuration at its FULL per-GPU batch: the instances (LV, SEIR: all of them; Robertson: every second; network100: every fourth) must equal
  the CPU oracle bit for bit (a tail-wave or arena-stride bug would not), and size-independent properties must
  hold on ALL instances (status 0, finite, conservation laws of the model);
* the device path against the digits printed in the reference's notebook (not only the oracle);
* the trajectory arena: tiled re-integration == resident, long trajectories at default settings.
"""
import os

import numpy as np
import pytest

from tests.helpers import make_oracle, make_problem

pytestmark = pytest.mark.gpu

CMP = [0, 1, 2, 3, 4, 5, 6, 7, 8]
CMP_B = [0, 1, 2, 3, 4, 5, 6, 9, 10, 12]


def _solver(name, rtol, atol, **kw):
    from sunode_amd.solver import AdjointSolver
    return AdjointSolver(make_problem(name), abstol=atol, reltol=rtol, backward_abstol=atol, backward_reltol=rtol,
                         quad_abstol=atol, quad_reltol=rtol, **kw)


def _oracle_sample(name, rtol, atol, b, idx):
    orc = make_oracle(name)
    cfg = orc.config(rtol=rtol, atol=atol, rtolB=rtol, atolB=atol, rtolQB=rtol, atolQB=atol)
    pr = b["pr"][idx] if b["rem_stride"] else b["pr"]
    if b["rem_stride"] == 0 and make_problem(name).n_remainder == 0:
        pr = np.zeros(0)
    y, st, sf = orc.solve_forward(cfg, b["y0"][idx], b["ps"][idx], pr, 0.0, b["tvals"], nthreads=os.cpu_count() or 8)
    g, lam, st2, sb = orc.solve_backward(cfg, b["tvals"][-1], 0.0, b["tvals"], b["grads"], nthreads=os.cpu_count() or 8)
    assert (st == 0).all() and (st2 == 0).all()
    return y, g, lam, sf, sb


@pytest.mark.parametrize("name,arena_gib", [("lv", None), ("robertson", None), ("seir", None), ("network100", None)])
def test_full_size_batches(name, arena_gib):
    import bench
    w = bench.WORKLOADS[name]
    prob = make_problem(name)
    B = w["batch"]
    b = bench.make_batch(name, prob, B)
    rt, at = w["rtol"], w["atol"]
    sol = _solver(name, rt, at)
    n_rem = prob.n_remainder
    pr_user = b["pr"][..., :n_rem] if n_rem else np.zeros(0)       # the API appends the hoisted values itself
    y, st, sf = sol.solve_forward_batch(0.0, b["tvals"], b["y0"], b["ps"], pr_user)
    g, lam, stb, sb = sol.solve_backward_batch(b["tvals"][-1], 0.0, b["tvals"], b["grads"])
    # properties on ALL instances
    assert (st == 0).all() and (stb == 0).all()
    assert np.isfinite(y).all() and np.isfinite(g).all() and np.isfinite(lam).all()
    assert (sf[:, 8] == sf[:, 0] + 1).all()                        # one stored point per step + the initial one
    if name == "robertson":                                        # mass conservation y1 + y2 + y3 = 1
        assert np.abs(y.sum(axis=2) - 1.0).max() < 5e-7
        assert (y > -1e-9).all()
    if name == "seir":                                             # group populations S+E+I+R are constant
        pop = y.reshape(B, -1, 4, 4).sum(axis=2)
        assert np.abs(pop / pop[:, :1] - 1.0).max() < 1e-7
    if name == "lv":
        assert (y > 0).all()
    # strided sample against the oracle, bit for bit (includes the last instance: tail of the last wave)
    # (LV and SEIR: EVERY instance of the batch; Robertson: every second (131 073); network100: every fourth (257) --
    # the oracle needs 5-15 s for these on the box's 16 cores, 40 s for network100; SA_FULLSIZE_STRIDE overrides)
    stride = int(os.environ.get("SA_FULLSIZE_STRIDE", {"lv": "1", "seir": "1", "robertson": "2"}.get(name, "4")))
    idx = np.unique(np.concatenate([np.arange(0, B, stride), [B - 1]]))
    yo, go, lo, sfo, sbo = _oracle_sample(name, rt, at, b, idx)
    np.testing.assert_array_equal(sf[idx][:, CMP], sfo[:, CMP])
    np.testing.assert_array_equal(sb[idx][:, CMP_B], sbo[:, CMP_B])
    np.testing.assert_array_equal(y[idx], yo)
    np.testing.assert_array_equal(g[idx], go)
    np.testing.assert_array_equal(lam[idx], lo)


def test_device_reproduces_the_reference_notebook_digits():
    """notebooks/from_sympy.ipynb cells 2, 8-12 THROUGH THE DEVICE: loss 185.95454144, dL/db, dL/dd with the
    notebook's seed-42 inputs (the analytic solution of the linear problem; printed at :240-242)."""
    from sunode_amd.solver import AdjointSolver
    prob = make_problem("notebook")
    rng = np.random.RandomState(42)
    b = rng.randn(2)
    dd = rng.randn(3)
    tvals = np.arange(20) / 100
    y0 = np.concatenate([np.arange(3.0) + dd[0] ** 2, b ** 3])
    f = np.linspace(0, 1, 50)
    for tol, rtol_val, rtol_grad in [(1e-10, 5e-9, 5e-8), (1e-13, 1e-10, 2e-9)]:
        sol = AdjointSolver(prob, abstol=tol, reltol=tol, backward_abstol=tol, backward_reltol=tol,
                            quad_abstol=tol, quad_reltol=tol)
        y, st, _ = sol.solve_forward_batch(0.0, tvals, y0[None], dd[None], f)
        assert st[0] == 0
        val = (y[0] ** 2).sum()
        g, lam, st2, _ = sol.solve_backward_batch(tvals[-1], 0.0, tvals, 2 * y)
        assert st2[0] == 0
        dy0 = -lam[0]
        grad_b = dy0[3:] * 3 * b ** 2
        grad_d = g[0].copy()
        grad_d[0] += dy0[:3].sum() * 2 * dd[0]
        np.testing.assert_allclose(val, 185.95454144, rtol=rtol_val)
        np.testing.assert_allclose(grad_b, [12.06638293, 0.86567236], rtol=rtol_grad)
        np.testing.assert_allclose(grad_d, [252.23687613, 12.10402814, 21.63579496], rtol=rtol_grad)
    # the same call through the scalar reference API (solve_forward / solve_backward, B = 1)
    sol.set_derivative_params(dd.view(sol.derivative_params_dtype)[0])
    sol.set_remaining_params(f.view(sol.remainder_params_dtype)[0])
    y_out, grad_out, lamda_out = sol.make_output_buffers(tvals)
    sol.solve_forward(0.0, tvals, y0, y_out)
    sol.solve_backward(tvals[-1], 0.0, tvals, 2 * y_out, grad_out, lamda_out)
    np.testing.assert_array_equal(y_out, y[0])
    np.testing.assert_array_equal(grad_out, g[0])


@pytest.mark.parametrize("name,B,small", [("robertson", 200, 0.005), ("lv", 300, 2.5e-3), ("seir", 70, 0.0035)])   # (compact records: robertson 40 B, seir 144 B per point)
def test_tiled_reintegration_equals_resident_arena(name, B, small):
    """A trajectory arena too small for the batch: the backward call re-integrates tile by tile (CVODES'
    check-point scheme).  Everything -- states, gradients, statuses, counters -- must equal the resident run."""
    import bench
    w = bench.WORKLOADS[name]
    prob = make_problem(name)
    b = bench.make_batch(name, prob, B)
    n_rem = prob.n_remainder
    pr_user = b["pr"][..., :n_rem] if n_rem else np.zeros(0)
    outs = []
    for arena_gib in (None, small):          # a few MB (at least one 64-instance group): several tiles
        sol = _solver(name, w["rtol"], w["atol"], arena_gib=arena_gib)
        y, st, sf = sol.solve_forward_batch(0.0, b["tvals"], b["y0"], b["ps"], pr_user)
        g, lam, stb, sb = sol.solve_backward_batch(b["tvals"][-1], 0.0, b["tvals"], b["grads"])
        # a second pass on the same handle must give the same answer; by then the handle has learnt how many rows
        # the batch needs (the first call starts with 512 and falls back to counting + re-integration)
        y2, st2, _ = sol.solve_forward_batch(0.0, b["tvals"], b["y0"], b["ps"], pr_user)
        g2, lam2, stb2, _ = sol.solve_backward_batch(b["tvals"][-1], 0.0, b["tvals"], b["grads"])
        np.testing.assert_array_equal(y2, y)
        np.testing.assert_array_equal(g2, g)
        info = sol._engine().arena_info()
        outs.append((y, st, sf, g, lam, stb, sb, info))
    res, til = outs
    assert not res[7][2] and til[7][2] and til[7][1] >= 2, (res[7], til[7])      # resident vs tiled, >= 2 tiles
    assert til[7][0] <= small * 2**30
    assert (res[1] == 0).all() and (res[5] == 0).all()
    for k in (0, 1, 3, 4, 5):
        np.testing.assert_array_equal(til[k], res[k])
    np.testing.assert_array_equal(til[2][:, CMP], res[2][:, CMP])
    np.testing.assert_array_equal(til[6][:, CMP_B], res[6][:, CMP_B])


def test_six_thousand_step_instance_at_default_settings():
    """A forward solve of > 6000 steps with the DEFAULT AdjointSolver settings (the reference's
    checkpoint_n = 500 000 just succeeds, /root/reference/sunode/solver.py:533,588): no max_steps tuning, same
    answer as the oracle."""
    from sunode_amd.solver import AdjointSolver
    prob = make_problem("lv")
    tv = np.linspace(0.0, 720.0, 13)
    y0 = np.array([[1.0, 0.1], [0.8, 0.3]])
    ps = np.array([[0.1, 0.2], [0.12, 0.18]])
    pr = np.array([[0.3, 0.4], [0.3, 0.4]])
    sol = AdjointSolver(prob)                                   # defaults: 1e-10 everywhere
    y, st, sf = sol.solve_forward_batch(0.0, tv, y0, ps, pr)
    g, lam, stb, sb = sol.solve_backward_batch(tv[-1], 0.0, tv, np.ones((13, 2)))
    assert (st == 0).all() and (stb == 0).all()
    assert sf[:, 0].min() > 6000
    orc = make_oracle("lv")
    cfg = orc.config()
    yo, so, sfo = orc.solve_forward(cfg, y0, ps, pr, 0.0, tv)
    go, lo, sbo, _ = orc.solve_backward(cfg, tv[-1], 0.0, tv, np.ones((13, 2)))
    np.testing.assert_array_equal(sf[:, CMP], sfo[:, CMP])
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(g, go)
    np.testing.assert_array_equal(lam, lo)


def test_full_size_hermite_and_sensitivities_lv():
    """The round-2 register-kernel paths at the full LV batch (B = 65 536): Hermite interpolation and forward
    sensitivities; a strided 1-in-16 sample must equal the oracle bit for bit, every instance must be finite."""
    import bench
    from sunode_amd import _native
    from sunode_amd.solver import Solver
    w = bench.WORKLOADS["lv"]
    prob = make_problem("lv")
    B = w["batch"]
    b = bench.make_batch("lv", prob, B)
    rt, at = w["rtol"], w["atol"]
    idx = np.arange(0, B, 16)
    n_rem = prob.n_remainder
    pr_user = b["pr"][..., :n_rem] if n_rem else np.zeros(0)
    orc = make_oracle("lv")
    pr_o = b["pr"][idx] if b["rem_stride"] else b["pr"]
    # Hermite
    assert _native.kernel_variant(prob.native_source(), hermite=True) == ("bdf_kernels.hip", 1)
    sol = _solver("lv", rt, at, interpolation="hermite")
    y, st, sf = sol.solve_forward_batch(0.0, b["tvals"], b["y0"], b["ps"], pr_user)
    g, lam, stb, sb = sol.solve_backward_batch(b["tvals"][-1], 0.0, b["tvals"], b["grads"])
    assert (st == 0).all() and (stb == 0).all() and np.isfinite(g).all() and np.isfinite(lam).all()
    cfg = orc.config(rtol=rt, atol=at, rtolB=rt, atolB=at, rtolQB=rt, atolQB=at, hermite=True)
    yo, so, _ = orc.solve_forward(cfg, b["y0"][idx], b["ps"][idx], pr_o, 0.0, b["tvals"], nthreads=os.cpu_count() or 8)
    go, lo, sbo, _ = orc.solve_backward(cfg, b["tvals"][-1], 0.0, b["tvals"], b["grads"], nthreads=os.cpu_count() or 8)
    assert (so == 0).all() and (sbo == 0).all()
    np.testing.assert_array_equal(y[idx], yo)
    np.testing.assert_array_equal(g[idx], go)
    np.testing.assert_array_equal(lam[idx], lo)
    # forward sensitivities
    assert _native.kernel_variant(prob.native_source(), sens=True) == ("bdf_kernels.hip", 1)
    ssol = Solver(prob, abstol=at, reltol=rt, sens_mode="simultaneous")
    sens0 = np.zeros((prob.n_params, prob.n_states))
    ys, S, sts, _ = ssol.solve_sens_batch(0.0, b["tvals"], b["y0"], b["ps"], pr_user, sens0)
    assert (sts == 0).all() and np.isfinite(S).all()
    cfg = orc.config(rtol=rt, atol=at)
    yso, So, sso, _ = orc.solve_sens(cfg, b["y0"][idx], b["ps"][idx], pr_o, sens0, 0.0, b["tvals"], mode="simultaneous", nthreads=os.cpu_count() or 8)
    assert (sso == 0).all()
    np.testing.assert_array_equal(ys[idx], yso)
    np.testing.assert_array_equal(S[idx], So)


def test_bench_rank_under_an_rccl_process_group_of_one(tmp_path):
    """the N > 1 branch of bench.py on the one GPU there is -- `run_rank` with the product `GpuEngine`
    under a world-size-1 `nccl` (= RCCL) process group: init_process_group(device_id=...), the barriers, the device
    all-reduces of the elapsed time / failed count and destroy_process_group all execute, and the line it prints
    equals what the same rank computes without a group (same kernels, same shard)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch",
           "4096", "--no-cpu-baseline", "--no-extra-configs"]
    lines = {}
    for tag, extra in (("group", ["--force-dist"]), ("plain", [])):
        res = subprocess.run(cmd + extra, env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
        lines[tag] = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])   # (RCCL prints too)
    g, p = lines["group"], lines["plain"]
    assert g["n_gpus"] == 1 and g["config"]["failed_instances"] == 0 and g["value"] > 0
    for key in ("fwd_steps_mean", "bwd_steps_mean", "stored_points_mean"):
        assert g["work"][key] == p["work"][key]
