"""N > 1 path on CPU: world_size-2 gloo process group, instance sharding + host-side gather.

The per-rank solve is played by the CPU oracle here (test-only stand-in for the GPU engine):
what is under test is the sharding / ordering / gather logic of sunode_amd.parallel."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, interleaved, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from sunode_amd.parallel import solve_sharded
    from tests.helpers import make_oracle, make_problem
    from tools.problems import lv_batch
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = make_problem("lv")
    orc = make_oracle("lv")
    B = 37                                   # ragged: not divisible by the world size
    d = lv_batch(B)
    ps = d["params"][:, prob.params_subset.subset_index]
    pr = d["params"][:, prob.params_subset.remainder_index]
    cfg = orc.config(rtol=1e-8, atol=1e-8, rtolB=1e-8, atolB=1e-8, rtolQB=1e-8, atolQB=1e-8)

    def solve_local(y0, ps_l, pr_l):
        if len(y0) == 0:
            return np.zeros((0, 50, 2)), np.zeros((0, 2)), np.zeros((0, 2)), np.zeros(0, np.int32)
        y, st, _ = orc.solve_forward(cfg, y0, ps_l, pr_l, 0.0, d["tvals"])
        g, lam, st2, _ = orc.solve_backward(cfg, d["tvals"][-1], 0.0, d["tvals"], np.ones((50, 2)))
        return y, g, lam, (st | st2).astype(np.int32)

    outs = solve_sharded(solve_local, [d["y0"], ps, pr], B, rank, world, interleaved)
    if rank == 0:
        np.savez(out_path, y=outs[0], g=outs[1], lam=outs[2], st=outs[3])
    else:
        assert all(o is None for o in outs)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("interleaved", [False, True])
def test_two_rank_sharded_solve_equals_single_process(tmp_path, interleaved):
    import torch.multiprocessing as mp
    from tests.helpers import make_oracle, make_problem
    from tools.problems import lv_batch
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, _free_port(), interleaved, out), nprocs=2, join=True)
    got = np.load(out)
    prob = make_problem("lv")
    orc = make_oracle("lv")
    d = lv_batch(37)
    ps = d["params"][:, prob.params_subset.subset_index]
    pr = d["params"][:, prob.params_subset.remainder_index]
    cfg = orc.config(rtol=1e-8, atol=1e-8, rtolB=1e-8, atolB=1e-8, rtolQB=1e-8, atolQB=1e-8)
    y, st, _ = orc.solve_forward(cfg, d["y0"], ps, pr, 0.0, d["tvals"])
    g, lam, st2, _ = orc.solve_backward(cfg, d["tvals"][-1], 0.0, d["tvals"], np.ones((50, 2)))
    np.testing.assert_array_equal(got["y"], y)
    np.testing.assert_array_equal(got["g"], g)
    np.testing.assert_array_equal(got["lam"], lam)
    assert (got["st"] == 0).all()


class _OracleEngine:
    """CPU stand-in for bench.GpuEngine (same interface): the rank's shard integrated by the oracle."""

    def __init__(self, name, prob, batch, tol, local_rank, arena_bytes=0):
        from tests.helpers import make_oracle
        self.orc = make_oracle(name)
        rt, at = tol
        self.cfg = self.orc.config(rtol=rt, atol=at, rtolB=rt, atolB=at, rtolQB=rt, atolQB=at)
        self.b = batch
        self.out = None

    def step(self):
        b = self.b
        y, st, sf = self.orc.solve_forward(self.cfg, b["y0"], b["ps"], b["pr"], 0.0, b["tvals"])
        g, lam, st2, sb = self.orc.solve_backward(self.cfg, b["tvals"][-1], 0.0, b["tvals"], b["grads"])
        self.out = (y, g, lam, st, st2, sf, sb)

    def kernel_ms(self):
        return 1.0, 1.0

    def sync(self):
        pass

    def results(self):
        y, g, lam, st, st2, sf, sb = self.out
        return dict(failed=int((st != 0).sum() + (st2 != 0).sum()), stats_f=sf.astype(float).mean(axis=0),
                    stats_b=sb.astype(float).mean(axis=0), arena=(0, 0, False), grad_sum=float(g.sum()))

    def close(self):
        pass


def _bench_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import json
    import bench
    args = bench.parse_args(["--gpus", str(world), "--steps", "1", "--warmup", "0", "--batch", "24"])
    out = bench.run_rank(args, backend="gloo", make_engine=_OracleEngine)
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump(out, fh)
    else:
        assert out is None


def test_bench_rank_function_two_ranks_gloo(tmp_path):
    """bench.py's own per-rank function (the code `python bench.py --gpus 2` runs under torch.distributed.run),
    world size 2 over gloo, the per-rank solve played by the CPU oracle: sharding through
    sunode_amd.parallel.shard_indices, barrier, max-reduce of the time, sum-reduce of the failed count."""
    import json
    import torch.multiprocessing as mp
    out = str(tmp_path / "bench.json")
    mp.spawn(_bench_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    line = json.load(open(out))
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 48 and line["config"]["batch_per_gpu"] == 24
    assert line["config"]["failed_instances"] == 0
    assert line["value"] > 0 and line["unit"] == "solves/s"
    assert abs(line["value"] - 48 * 1 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    # rank 0 integrated draws 0..23 of the global 48-draw batch (contiguous shards)
    import bench
    from tests.helpers import make_oracle, make_problem
    prob = make_problem("lv")
    full = bench.make_batch("lv", prob, 48)
    orc = make_oracle("lv")
    cfg = orc.config(rtol=1e-8, atol=1e-8, rtolB=1e-8, atolB=1e-8, rtolQB=1e-8, atolQB=1e-8)
    _, _, sf = orc.solve_forward(cfg, full["y0"][:24], full["ps"][:24], full["pr"][:24], 0.0, full["tvals"])
    assert line["work"]["fwd_steps_mean"] == float(sf[:, 0].mean())


def test_bench_single_process_launcher_two_devices(monkeypatch):
    """`bench.py --gpus 2 --single-process`: one process, one engine + host thread per device, no process group;
    the engines get the two contiguous halves of the global 2 x B batch (the per-device solve played by the CPU
    oracle), the line has the contract fields of the one-process-per-GPU launcher."""
    import bench
    from sunode_amd import _native
    monkeypatch.setattr(_native, "device_memory", lambda d: (200 << 30, 288 << 30))
    made = []

    import threading
    lock = threading.Lock()         # (the CPU oracle keeps the last forward solve in its handle: one step at a time)

    def factory(name, prob, batch, tol, device, arena_bytes=0):
        e = _OracleEngine(name, prob, batch, tol, device)
        step = e.step

        def locked_step():
            with lock:
                step()
                e.kept = e.out
        e.step = locked_step
        res = e.results
        e.results = lambda: (setattr(e, "out", e.kept), res())[1]
        made.append((device, arena_bytes, batch["y0"].shape[0], batch["ps"][0].copy()))
        return e
    args = bench.parse_args(["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "12", "--single-process",
                             "--devices", "0,0"])
    line = bench.run_single_process(args, make_engine=factory)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 24
    assert line["config"]["failed_instances"] == 0 and "one process" in line["config"]["parallelism"]
    assert abs(line["value"] - 24 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    from tests.helpers import make_problem
    full = bench.make_batch("lv", make_problem("lv"), 24)
    assert [m[2] for m in made] == [12, 12]
    np.testing.assert_array_equal(made[0][3], full["ps"][0])
    np.testing.assert_array_equal(made[1][3], full["ps"][12])
    assert [m[1] for m in made] == [48 << 30, 48 << 30]              # two engines on one device share its arena budget
    with pytest.raises(SystemExit):
        bench.run_single_process(bench.parse_args(["--gpus", "3", "--single-process", "--devices", "0,1"]))


def test_bench_gpus_flag_must_match_world(monkeypatch):
    import bench
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit):
        bench.run_rank(bench.parse_args(["--gpus", "2"]))


def test_shard_bounds_cover_and_balance():
    from sunode_amd.parallel import shard_bounds, shard_indices
    for n in (0, 1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
            inter = np.concatenate([shard_indices(n, r, world, True) for r in range(world)])
            assert sorted(inter.tolist()) == list(range(n))
