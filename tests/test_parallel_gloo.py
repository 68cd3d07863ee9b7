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
