"""One Python process driving several GPUs: ``AdjointSolver(problem, devices=[...])`` / ``Solver(..., devices=[...])``
(the reference's call pattern is one solver object inside one PyMC process,
/root/reference/sunode/wrappers/as_pytensor.py:279-344).

CPU: the sharding / threading / arena-split logic against a recording fake of the native handle.
GPU (-m gpu, one device on the box): ``devices=[0, 0]`` -- two handles, two streams, two arenas on one device -- must
equal the one-handle result bit for bit, through the batch API and under the batched pytensor Op's solver calls.
"""
import threading

import numpy as np
import pytest

from tests.helpers import make_problem


class FakeNative:
    """Records what a NativeSolver would be asked to do and fills the outputs with values that identify the instance
    (so a row written by the wrong shard, or to the wrong place, shows)."""
    created = []
    N_STATS = 16
    rendezvous = None           # threading.Barrier: every handle's forward call must be in flight at the same time

    def __init__(self, source, *, device=0, arena_bytes=0, **kw):
        self.device, self.arena_bytes, self.kw = device, arena_bytes, kw
        self._opt_kw = dict(max_retries_fwd=5, max_retries_bwd=50)
        self.calls = []
        FakeNative.created.append(self)

    def set_options(self, **kw):
        self._opt_kw.update(kw)

    def solve(self, mem, B, y0, ps, pr, rem_stride, t0, tvals, n_t, y_out, status, stats, adjoint=False):
        self.calls.append(("forward" if adjoint else "solve", B, threading.get_ident()))
        if FakeNative.rendezvous is not None:
            FakeNative.rendezvous.wait(timeout=30)
        assert y0.shape[0] == B and y_out.shape[0] == B and status.shape == (B,) and y_out.flags["C_CONTIGUOUS"]
        self.last_ps = np.array(ps[:B]) if np.ndim(ps) == 2 else None
        y_out[...] = y0[:, None, :] * (1.0 + np.asarray(tvals)[None, :, None]) + (ps[:, :1, None] if np.ndim(ps) == 2 else 0.0)
        status[...] = 0
        stats[:, 0] = 100 + self.device

    def solve_backward(self, mem, B, ps, pr, rem_stride, t0, tend, tvals, n_t, grads, gstride, grad_out, lamda_out,
                       status, stats, lamda_all=None, quad_all=None):
        self.calls.append(("backward", B, threading.get_ident()))
        assert grad_out.shape[0] == B and lamda_out.shape[0] == B
        np.testing.assert_array_equal(ps[:B], self.last_ps)          # the shard this handle integrated forward
        g = grads if gstride == 0 else grads.sum(axis=(1, 2))[:, None]
        grad_out[...] = ps[:B, :grad_out.shape[1]] * 2.0 + (0.0 if gstride == 0 else g)
        lamda_out[...] = -1.0 - ps[:B, :1]
        status[...] = 0
        if lamda_all is not None:
            lamda_all[...] = 7.0
            quad_all[...] = 8.0

    def solve_sens(self, mem, ism, scaling, B, y0, ps, pr, rem_stride, sens0, t0, tvals, n_t, y_out, sens_out,
                   status, stats):
        self.calls.append(("sens", B, threading.get_ident()))
        y_out[...] = y0[:, None, :]
        sens_out[...] = ps[:, None, :, None]
        status[...] = 0

    def last_kernel_ms(self):
        return 1.0 + self.device, 2.0 + self.device


@pytest.fixture
def fake_native(monkeypatch):
    from sunode_amd import _native
    FakeNative.created = []
    monkeypatch.setattr(_native, "NativeSolver", FakeNative)
    monkeypatch.setattr(_native, "device_memory", lambda d: (200 << 30, 288 << 30))
    return FakeNative


def _lv_inputs(B):
    from tools.problems import lv_batch
    d = lv_batch(B)
    return d, d["params"][:, :2], d["params"][:, 2:]


def test_shards_are_contiguous_balanced_and_land_in_the_callers_rows(fake_native):
    from sunode_amd.solver import AdjointSolver
    prob = make_problem("lv")
    B = 37
    d, ps, pr = _lv_inputs(B)
    sol = AdjointSolver(prob, devices=[0, 1, 2, 3], arena_gib=8)
    fake_native.rendezvous = threading.Barrier(4)       # passes only if the four calls overlap in time
    try:
        y, st, stats = sol.solve_forward_batch(0.0, d["tvals"], d["y0"], ps, pr)
    finally:
        fake_native.rendezvous = None
    handles = fake_native.created
    assert [h.device for h in handles] == [0, 1, 2, 3]
    assert [h.arena_bytes for h in handles] == [8 << 30] * 4            # one handle per device: the whole budget each
    assert [h.calls[0][1] for h in handles] == [10, 9, 9, 9]            # balanced contiguous shards
    assert len({h.calls[0][2] for h in handles}) == 4                    # one host thread per handle
    want = d["y0"][:, None, :] * (1.0 + d["tvals"][None, :, None]) + ps[:, :1, None]
    np.testing.assert_array_equal(y, want)
    np.testing.assert_array_equal(stats[:, 0], np.repeat([100, 101, 102, 103], [10, 9, 9, 9]))
    grads = np.cos(np.arange(B * len(d["tvals"]) * 2.0)).reshape(B, len(d["tvals"]), 2)
    g, lam, stb, _ = sol.solve_backward_batch(d["tvals"][-1], 0.0, d["tvals"], grads)     # per-instance cotangents
    np.testing.assert_array_equal(g, ps * 2.0 + grads.sum(axis=(1, 2))[:, None])
    np.testing.assert_array_equal(lam, np.tile(-1.0 - ps[:, :1], (1, 2)))
    g2, lam2, _, _, la, qa = sol.solve_backward_batch(d["tvals"][-1], 0.0, d["tvals"], grads[0], return_all=True)
    np.testing.assert_array_equal(g2, ps * 2.0)                          # shared cotangent: passed whole to every handle
    assert (la == 7.0).all() and (qa == 8.0).all()
    assert sol.last_kernel_ms() == (4.0, 5.0)                            # the slowest handle's


def test_handles_sharing_a_device_split_its_arena_budget(fake_native):
    from sunode_amd.solver import AdjointSolver
    prob = make_problem("lv")
    d, ps, pr = _lv_inputs(5)
    sol = AdjointSolver(prob, devices=[0, 0, 1], arena_gib=10)
    sol.solve_forward_batch(0.0, d["tvals"], d["y0"], ps, pr)
    assert [h.arena_bytes for h in fake_native.created] == [5 << 30, 5 << 30, 10 << 30]
    fake_native.created.clear()
    sol = AdjointSolver(prob, devices=[0, 0])                           # no explicit budget: the library default
    sol.solve_forward_batch(0.0, d["tvals"], d["y0"], ps, pr)           # (96 GiB, at most 60 % of the free HBM) ONCE
    assert [h.arena_bytes for h in fake_native.created] == [48 << 30, 48 << 30]
    fake_native.created.clear()
    sol = AdjointSolver(prob)                                            # default: one handle, the library decides
    sol.solve_forward_batch(0.0, d["tvals"], d["y0"], ps, pr)
    assert [(h.device, h.arena_bytes) for h in fake_native.created] == [(0, 0)]
    assert len({c[2] for c in fake_native.created[0].calls}) == 1 and \
        fake_native.created[0].calls[0][2] == threading.get_ident()     # ... and no thread


def test_more_handles_than_instances_and_forward_solver(fake_native):
    from sunode_amd.solver import Solver
    prob = make_problem("lv")
    d, ps, pr = _lv_inputs(3)
    sol = Solver(prob, devices=[0, 1, 2, 3, 4, 5, 6, 7])
    y, st, _ = sol.solve_batch(0.0, d["tvals"], d["y0"], ps, pr)
    assert [len(h.calls) for h in fake_native.created] == [1, 1, 1, 0, 0, 0, 0, 0]      # empty shards are not launched
    np.testing.assert_array_equal(y, d["y0"][:, None, :] * (1.0 + d["tvals"][None, :, None]) + ps[:, :1, None])
    sol = Solver(prob, devices=[0, 1], sens_mode="simultaneous")
    y, S, st, _ = sol.solve_sens_batch(0.0, d["tvals"], d["y0"], ps, pr, np.zeros((2, 2)))
    np.testing.assert_array_equal(S, np.broadcast_to(ps[:, None, :, None], S.shape))
    with pytest.raises(ValueError):
        Solver(prob, devices=[])


def test_a_failing_shard_raises_after_the_others_finished(fake_native, monkeypatch):
    from sunode_amd.solver import AdjointSolver
    prob = make_problem("lv")
    d, ps, pr = _lv_inputs(8)
    done = []

    def solve(self, mem, B, *a, **k):
        if self.device == 1:
            raise RuntimeError("device 1 lost")
        done.append(self.device)
    monkeypatch.setattr(FakeNative, "solve", solve)
    sol = AdjointSolver(prob, devices=[0, 1, 2])
    with pytest.raises(RuntimeError, match="device 1 lost"):
        sol.solve_forward_batch(0.0, d["tvals"], d["y0"], ps, pr)
    assert sorted(done) == [0, 2]


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["lv", "seir"])
def test_two_handles_on_one_device_equal_one_handle(name):
    """devices=[0, 0]: two handles (two streams, two arenas, the device's budget split) driven from two host threads
    give the one-handle results bit for bit -- forward states, counters, gradients with per-instance cotangents."""
    from sunode_amd.solver import AdjointSolver
    from tools.problems import lv_batch, seir_batch
    prob = make_problem(name)
    B = 1000 if name == "lv" else 200
    if name == "lv":
        d = lv_batch(B); ps, pr = d["params"][:, :2], d["params"][:, 2:]
    else:
        d = seir_batch(B); ps, pr = d["ps"], d["pr"]
    tv = d["tvals"]
    n = prob.n_states
    rng = np.random.RandomState(3)
    grads = rng.randn(B, len(tv), n)
    res = {}
    for devs in ([0], [0, 0], [0, 0, 0]):
        sol = AdjointSolver(prob, abstol=1e-8, reltol=1e-8, backward_abstol=1e-8, backward_reltol=1e-8,
                            quad_abstol=1e-8, quad_reltol=1e-8, devices=devs, arena_gib=6)
        y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
        g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
        assert (st == 0).all() and (stb == 0).all()
        assert len(sol._engines()) == len(devs)
        assert [e._opt_kw["arena_bytes"] for e in sol._engines()] == [(6 << 30) // len(devs)] * len(devs)
        res[len(devs)] = (y, stats[:, :9], g, lam, statsb[:, :8])
    for k in (2, 3):
        for a, b in zip(res[1], res[k]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_forward_solver_and_sensitivities_on_two_handles():
    from sunode_amd.solver import Solver
    from tools.problems import lv_batch
    prob = make_problem("lv")
    d = lv_batch(300)
    ps, pr = d["params"][:, :2], d["params"][:, 2:]
    res = {}
    for devs in ([0], [0, 0]):
        sol = Solver(prob, abstol=1e-8, reltol=1e-8, sens_mode="staggered", devices=devs)
        res[len(devs)] = sol.solve_sens_batch(0.0, d["tvals"], d["y0"], ps, pr, np.zeros((2, 2)))
    for a, b in zip(res[1], res[2]):
        np.testing.assert_array_equal(a, b)
    from sunode_amd import _native
    assert _native.device_count() >= 1
    free_b, total_b = _native.device_memory(0)
    assert 0 < free_b <= total_b and total_b > (100 << 30)             # MI355X: 288 GB


def test_interleaved_shards_keep_the_callers_order(fake_native):
    """``interleaved=True``: handle r integrates the instances r, r + G, ...; inputs are gathered
    into handle-major order, every result comes back in the caller's order, and the backward call hands every handle
    the instances it integrated forward (the fake asserts that)."""
    from sunode_amd.solver import AdjointSolver, Solver
    prob = make_problem("lv")
    B = 11
    d, ps, pr = _lv_inputs(B)
    sol = AdjointSolver(prob, devices=[0, 1, 2], interleaved=True, arena_gib=3)
    y, st, stats = sol.solve_forward_batch(0.0, d["tvals"], d["y0"], ps, pr)
    handles = fake_native.created
    assert [h.calls[0][1] for h in handles] == [4, 4, 3]                 # instances 0,3,6,9 | 1,4,7,10 | 2,5,8
    np.testing.assert_array_equal(handles[1].last_ps, ps[1::3])
    np.testing.assert_array_equal(y, d["y0"][:, None, :] * (1.0 + d["tvals"][None, :, None]) + ps[:, :1, None])
    np.testing.assert_array_equal(stats[:, 0], 100 + np.arange(B) % 3)  # which handle integrated which instance
    grads = np.cos(np.arange(B * len(d["tvals"]) * 2.0)).reshape(B, len(d["tvals"]), 2)
    g, lam, stb, _ = sol.solve_backward_batch(d["tvals"][-1], 0.0, d["tvals"], grads)
    np.testing.assert_array_equal(g, ps * 2.0 + grads.sum(axis=(1, 2))[:, None])
    np.testing.assert_array_equal(lam, np.tile(-1.0 - ps[:, :1], (1, 2)))
    g2, _, _, _, la, qa = sol.solve_backward_batch(d["tvals"][-1], 0.0, d["tvals"], grads[0], return_all=True)
    np.testing.assert_array_equal(g2, ps * 2.0)
    assert la.shape == (B, len(d["tvals"]), 2) and (la == 7.0).all() and (qa == 8.0).all()
    fake_native.created.clear()
    fwd = Solver(prob, devices=[0, 1], interleaved=True, sens_mode="simultaneous")
    y, S, st, _ = fwd.solve_sens_batch(0.0, d["tvals"], d["y0"], ps, pr, np.zeros((2, 2)))
    np.testing.assert_array_equal(S, np.broadcast_to(ps[:, None, :, None], S.shape))
    np.testing.assert_array_equal(y, np.broadcast_to(d["y0"][:, None, :], y.shape))
    one = Solver(prob, devices=[0], interleaved=True)                    # one handle: nothing to reorder
    y1, _, _ = one.solve_batch(0.0, d["tvals"], d["y0"], ps, pr)
    np.testing.assert_array_equal(y1, d["y0"][:, None, :] * (1.0 + d["tvals"][None, :, None]) + ps[:, :1, None])


@pytest.mark.gpu
@pytest.mark.parametrize("name,B", [("seir", 8 * 16384), ("network100", 8 * 64)])
def test_eight_handles_carry_the_global_batch_of_an_eight_gpu_node(name, B):
    """8-GPU readiness on the one-GPU box.  BASELINE config 4's GLOBAL batch (131 072 SEIR instances =
    8 x 16 384) -- and 8 shards of the 100-state network -- through EIGHT handles on device 0 (eight streams, eight
    arenas sharing the device's budget eight ways, eight host threads), contiguous and interleaved shards: statuses,
    counters, states and gradients equal the one-handle run bit for bit."""
    from sunode_amd.solver import AdjointSolver
    from tools.problems import network_batch, seir_batch
    prob = make_problem(name)
    d = seir_batch(B) if name == "seir" else network_batch(B)
    tv = d["tvals"]
    k = np.arange(len(tv))[:, None]; i = np.arange(prob.n_states)[None, :]
    grads = 1.0 + 0.5 * np.cos(1.7 * k + 0.9 * i)
    tol = dict(abstol=1e-8, reltol=1e-8, backward_abstol=1e-8, backward_reltol=1e-8, quad_abstol=1e-8, quad_reltol=1e-8)
    ref = None
    for devs, inter in (([0], False), ([0] * 8, False), ([0] * 8, True)):
        sol = AdjointSolver(prob, devices=devs, interleaved=inter, arena_gib=64, **tol)
        y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], d["ps"], d["pr"])
        g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
        assert (st == 0).all() and (stb == 0).all()
        assert [e._opt_kw["arena_bytes"] for e in sol._engines()] == [(64 << 30) // len(devs)] * len(devs)
        got = (y.copy(), stats[:, :9].copy(), g.copy(), lam.copy(), statsb[:, :8].copy())
        if ref is None:
            ref = got
        else:
            for a, b in zip(ref, got):
                np.testing.assert_array_equal(a, b)
        del sol


def test_caller_allocated_outputs_and_the_opt_in_pool(fake_native):
    """``out=`` on the batch API (the reference's convention: the caller allocates, the solver writes in place,
    /root/reference/sunode/solver.py:682,723-724) and the lifetime rules of the results: by default every call
    returns FRESH arrays (a result kept only through a raw pointer survives the next call); ``reuse_outputs=True`` is
    the explicit opt-in to recycled arrays."""
    import ctypes
    from sunode_amd.solver import AdjointSolver, Solver
    prob = make_problem("lv")
    B = 11
    d, ps, pr = _lv_inputs(B)
    tv = d["tvals"]
    want = d["y0"][:, None, :] * (1.0 + tv[None, :, None]) + ps[:, :1, None]
    for devices, interleaved in (([0], False), ([0, 1, 2], False), ([0, 1, 2], True)):
        sol = AdjointSolver(prob, devices=devices, interleaved=interleaved)
        y_buf, st_buf = np.full((B, len(tv), 2), -1.0), np.full(B, 77, np.int32)
        y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr, out={"y_out": y_buf, "status": st_buf})
        assert y is y_buf and st is st_buf and stats.shape == (B, 16)
        np.testing.assert_array_equal(y_buf, want)
        assert (st_buf == 0).all()
        g_buf, l_buf = np.empty((B, 2)), np.empty((B, 2))
        g, lam, stb, _ = sol.solve_backward_batch(tv[-1], 0.0, tv, np.ones((len(tv), 2)), out=(g_buf, l_buf))
        assert g is g_buf and lam is l_buf
        np.testing.assert_array_equal(g_buf, ps * 2.0)
        # default lifetime: a result held only through a raw pointer is not overwritten by the next call
        y1, _, _ = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
        addr = y1.ctypes.data
        keep = y1                                    # (keeps the memory alive; the solver cannot know about `addr`)
        y2, _, _ = sol.solve_forward_batch(0.0, tv, 2.0 * d["y0"], ps, pr)
        assert y2.ctypes.data != addr
        seen = np.ctypeslib.as_array(ctypes.cast(addr, ctypes.POINTER(ctypes.c_double)), shape=y1.shape)
        np.testing.assert_array_equal(seen, want)
        del keep
    # wrong shape / dtype / layout / read-only: refused before anything runs
    sol = AdjointSolver(prob)
    for bad in (np.empty((B, len(tv), 3)), np.empty((B, len(tv), 2), np.float32), np.empty((len(tv), B, 2)).transpose(1, 0, 2)):
        with pytest.raises(ValueError):
            sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr, out=[bad])
    ro = np.empty((B, len(tv), 2)); ro.setflags(write=False)
    with pytest.raises(ValueError):
        sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr, out=[ro])
    with pytest.raises(ValueError):
        sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr, out={"y": np.empty((B, len(tv), 2))})
    sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
    with pytest.raises(ValueError):
        sol.solve_backward_batch(tv[-1], 0.0, tv, np.ones((len(tv), 2)), out={"lamda_all": np.empty((B, len(tv), 2))})
    # the opt-in pool: same array again, contract = valid until the next call of the same method
    sol = AdjointSolver(prob, reuse_outputs=True)
    y1, _, _ = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
    y2, _, _ = sol.solve_forward_batch(0.0, tv, 2.0 * d["y0"], ps, pr)
    assert y2 is y1
    plain = Solver(prob)
    y_buf = np.empty((B, len(tv), 2))
    y, st, _ = plain.solve_batch(0.0, tv, d["y0"], ps, pr, out=[y_buf])
    assert y is y_buf and (st == 0).all()
    sens = Solver(prob, sens_mode="simultaneous")
    s_buf = np.empty((B, len(tv), 2, 2))
    _, s, _, _ = sens.solve_sens_batch(0.0, tv, d["y0"], ps, pr, np.zeros((2, 2)), out={"sens_out": s_buf})
    assert s is s_buf


def test_empty_time_grid_returns_zeroed_status_and_counters(fake_native, monkeypatch):
    """The library returns early for n_t == 0 without writing anything: status / counters must not be uninitialised
    memory."""
    from sunode_amd.solver import AdjointSolver

    def untouched(self, mem, B, y0, ps, pr, rem_stride, t0, tvals, n_t, y_out, status, stats, adjoint=False):
        pass
    monkeypatch.setattr(fake_native, "solve", untouched)
    prob = make_problem("lv")
    d, ps, pr = _lv_inputs(6)
    for reuse in (False, True):
        sol = AdjointSolver(prob, reuse_outputs=reuse)
        for _ in range(3):
            y, st, stats = sol.solve_forward_batch(0.0, np.zeros(0), d["y0"], ps, pr)
            assert y.shape == (6, 0, 2) and (st == 0).all() and (stats == 0).all()
            st[:] = 5; stats[:] = 9                   # (a recycled array shows this again unless it is re-zeroed)


def test_small_batch_mapping_selection(fake_native, monkeypatch):
    """Which handles a batch runs on (AdjointSolver._select_mapping): 4-lane groups for a 4- / 5-state model the engine
    maps to one lane, while a HANDLE's share of the batch is <= 16 384; never for 2 / 3 states, never under
    SA_FORCE_GROUP, SA_BATCH_MAPPING=fixed or batch_mapping="fixed"; the backward call runs on the forward call's
    handles whatever happens in between."""
    from sunode_amd import _native
    from sunode_amd.solver import AdjointSolver
    monkeypatch.delenv("SA_FORCE_GROUP", raising=False)
    monkeypatch.delenv("SA_BATCH_MAPPING", raising=False)
    nb = make_problem("notebook")                       # 5 states, 3 differentiated parameters
    src = nb.native_source()
    assert _native.kernel_variant(src) == ("bdf_kernels.hip", 1)
    assert [_native.small_batch_group(src, batch=b) for b in (1, 4096, 4097, 8192, 8193, 16384, 16385)] == \
        ["wave16", "wave16", "wave8", "wave8", "wave4", "wave4", None]
    assert _native.kernel_variant(src, group="wave4") == ("bdf_wave.hip", 4)
    assert _native.code_object_path(src, compact=True, group="wave4") != _native.code_object_path(src, compact=True)
    assert _native.small_batch_group(make_problem("lv").native_source()) is None
    assert _native.small_batch_group(make_problem("robertson").native_source()) is None
    seir = make_problem("seir").native_source()                     # 4-lane groups by default: 16 / 8 lanes for small batches
    assert [_native.small_batch_group(seir, batch=b) for b in (1, 4096, 4097, 8192, 8193, 16384)] == \
        ["wave16", "wave16", "wave8", "wave8", None, None]
    assert _native.small_batch_group(make_problem("network24").native_source(), batch=8) is None    # (not measured: unchanged)
    monkeypatch.setenv("SA_FORCE_GROUP", "1")
    assert _native.small_batch_group(src) is None
    monkeypatch.delenv("SA_FORCE_GROUP")
    monkeypatch.setenv("SA_BATCH_MAPPING", "fixed")
    assert _native.small_batch_group(src) is None
    monkeypatch.delenv("SA_BATCH_MAPPING")

    def run(sol, B):
        y0 = np.ones((B, 5)); ps = np.ones((B, 3)); pr = np.linspace(0, 1, 50); tv = np.arange(4) / 10
        sol.solve_forward_batch(0.0, tv, y0, ps, pr)
        return [h.kw.get("group") for h in fake_native.created]
    sol = AdjointSolver(nb)
    assert run(sol, 100) == ["wave16"]
    assert run(sol, 4096) == ["wave16"]                                  # same handle again
    assert run(sol, 16384) == ["wave16", "wave4"]
    assert run(sol, 16385) == ["wave16", "wave4", None]                  # the one-lane handle appears
    fwd_handle = fake_native.created[2]
    assert run(sol, 10) == ["wave16", "wave4", None]
    sol.solve_forward_batch(0.0, np.arange(4) / 10, np.ones((20000, 5)), np.ones((20000, 3)), np.linspace(0, 1, 50))
    sol.solve_backward_batch(0.3, 0.0, np.arange(4) / 10, np.ones((4, 5)))
    assert fwd_handle.calls[-1][0] == "backward" and fwd_handle.calls[-2][0] == "forward"
    fake_native.created.clear()
    two = AdjointSolver(nb, devices=[0, 1])
    assert run(two, 30000) == ["wave4", "wave4"]                         # 15 000 per handle
    assert run(two, 40000) == ["wave4", "wave4", None, None]
    assert run(two, 8000) == ["wave4", "wave4", None, None, "wave16", "wave16"]
    fake_native.created.clear()
    assert run(AdjointSolver(nb, batch_mapping="fixed"), 100) == [None]
    assert _native.small_batch_group(src) == "wave16"
    with pytest.raises(ValueError):
        AdjointSolver(nb, batch_mapping="sometimes")
