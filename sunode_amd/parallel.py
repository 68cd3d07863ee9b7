"""Batch sharding across the GPUs of one node (one process per GPU).

The path shards embarrassingly by instance: every rank integrates its
own slice of the parameter draws on its own GPU with its own ``sa_solver`` handle and
trajectory arena; there is no exchange step, hence no data-path collective.  The only
communication is the optional host-side gather of the per-instance results to rank 0
(``torch.distributed.gather`` on CPU tensors -- works with the ``gloo`` backend, or with
RCCL on device tensors) and the barrier / max-reduce used for timing in bench.py.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced partition: rank r owns [lo, hi); sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_indices(n_items: int, rank: int, world: int, interleaved: bool = False) -> np.ndarray:
    """Instance indices of one rank.  ``interleaved`` (i mod world == rank) balances batches whose
    cost varies systematically with the index (e.g. sorted stiffness); output order stays global."""
    if interleaved:
        return np.arange(rank, n_items, world)
    lo, hi = shard_bounds(n_items, rank, world)
    return np.arange(lo, hi)


def gather_to_root(local: np.ndarray, n_items: int, rank: int, world: int, interleaved: bool = False,
                   group=None) -> Optional[np.ndarray]:
    """Host-side gather of per-instance results (leading axis = local instances) to rank 0.

    Returns the full array (leading axis ``n_items``, global instance order) on rank 0 and None
    elsewhere.  With world == 1 no process group is needed."""
    local = np.ascontiguousarray(local)
    if world == 1:
        return local
    import torch
    import torch.distributed as dist
    counts = [len(shard_indices(n_items, r, world, interleaved)) for r in range(world)]
    pad = max(counts)
    buf = np.zeros((pad,) + local.shape[1:], dtype=local.dtype)
    buf[:local.shape[0]] = local
    t = torch.from_numpy(buf)
    gathered = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, gathered, dst=0, group=group)
    if rank != 0:
        return None
    out = np.zeros((n_items,) + local.shape[1:], dtype=local.dtype)
    for r in range(world):
        idx = shard_indices(n_items, r, world, interleaved)
        out[idx] = gathered[r].numpy()[:len(idx)]
    return out


def solve_sharded(solve_local, arrays: Sequence[np.ndarray], n_items: int, rank: int, world: int,
                  interleaved: bool = False, group=None):
    """Run ``solve_local(*local_arrays) -> tuple of per-instance arrays`` on this rank's shard and
    gather every output to rank 0.  ``arrays`` are full-batch inputs (leading axis n_items)."""
    idx = shard_indices(n_items, rank, world, interleaved)
    outs = solve_local(*[np.asarray(a)[idx] for a in arrays])
    return tuple(gather_to_root(o, n_items, rank, world, interleaved, group) for o in outs)
