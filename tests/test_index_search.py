"""The interpolation index without CVAfindIndex' walk (csrc/bdf_kernels.hip search_left / search_right, round 6).

CVODES' adjoint interpolation walks from the last bracketing index one stored point at a time
(`CVAfindIndex`; the wrappers call it from every backward callback, /root/reference/sunode/solver.py:723-784 drives it
through CVodeB).  The device finds the same index by galloping + section search.  The argument: the stored times increase
strictly, so the walk's comparisons are monotone in the index -- checked here on a host transcription of BOTH algorithms
(no GPU; the device code itself is compared with the oracle's plain walk by the -m gpu parity tests, bit for bit)."""
import numpy as np
import pytest

NP = 4          # SA_SEARCH_PROBES


def walk_left(T, t, indx, tprev, tcur):
    """bdf_kernels.hip interp_y, plain walk (= oracle/cvodes_oracle.c): returns indx, t[indx-1], t[indx]"""
    while True:
        if indx == 0:
            break
        if (t - tprev) <= 0.0:
            indx -= 1
            tcur = tprev
            if indx > 0:
                tprev = T[indx - 1]
        else:
            break
    return indx, tprev, tcur


def walk_right(T, t, indx, tprev, tcur):
    last = len(T) - 1
    while True:
        if indx >= last:
            break
        if (t - tcur) > 0.0:
            indx += 1
            tprev = tcur
            tcur = T[indx]
        else:
            break
    return indx, tprev, tcur


def search_left(T, t, hi, thv):
    """first k in [0, hi] with (t - T[k]) <= 0, given that it holds at hi; returns k, T[k-1] (k > 0), T[k], probes"""
    lo, step, tlv, probes = -1, 2, None, 0
    while lo < 0 and hi > 0:
        ks = [max(hi - (step << j), 0) for j in range(NP)]
        vs = [T[k] for k in ks]; probes += NP
        for k, v in zip(ks, vs):                  # nearest first
            if (t - v) <= 0.0:
                hi, thv = k, v
            else:
                lo, tlv = k, v
                break
        step <<= NP
    while hi - lo > 1:
        n, lo0, hi0 = hi - lo, lo, hi
        ks = [min(max(lo0 + (n * (j + 1)) // (NP + 1), lo0 + 1), hi0 - 1) for j in range(NP)]
        vs = [T[k] for k in ks]; probes += NP
        for k, v in zip(ks, vs):                  # ascending
            if (t - v) <= 0.0:
                hi, thv = k, v
                break
            lo, tlv = k, v
    return hi, tlv, thv, probes


def search_right(T, t, lo, tlv):
    """first k in (lo, last] with (t - T[k]) <= 0, or last; (t - T[lo]) > 0 on entry"""
    last = len(T) - 1
    hi, step, thv, ran_off, probes = -1, 1, None, False, 0
    while hi < 0:
        ks = [min(lo + (step << j), last) for j in range(NP)]
        vs = [T[k] for k in ks]; probes += NP
        for k, v in zip(ks, vs):
            if (t - v) > 0.0:
                lo, tlv = k, v
                if k == last:
                    hi, thv, ran_off = last, v, True
                    break
            else:
                hi, thv = k, v
                break
        step <<= NP
    if ran_off:
        lo, tlv = last - 1, T[last - 1]
    while hi - lo > 1:
        n, lo0, hi0 = hi - lo, lo, hi
        ks = [min(max(lo0 + (n * (j + 1)) // (NP + 1), lo0 + 1), hi0 - 1) for j in range(NP)]
        vs = [T[k] for k in ks]; probes += NP
        for k, v in zip(ks, vs):
            if (t - v) > 0.0:
                lo, tlv = k, v
            else:
                hi, thv = k, v
                break
    return hi, tlv, thv, probes


def grids():
    rng = np.random.default_rng(20260930)
    for n in (2, 3, 7, 64, 1000):
        for kind in ("uniform", "geometric", "random"):
            if kind == "uniform":
                T = np.linspace(0.0, 10.0, n)
            elif kind == "geometric":             # Robertson-like: step sizes over ten decades
                T = np.r_[0.0, np.cumsum(1e-6 * 1.03 ** np.arange(n - 1))]
            else:
                T = np.r_[0.0, np.cumsum(rng.uniform(1e-9, 1.0, n - 1))]
            yield T, rng


def test_left_search_ends_where_the_walk_ends():
    for T, rng in grids():
        n = len(T)
        for _ in range(400):
            ilast = int(rng.integers(1, n))
            # t somewhere left of the bracket (t[ilast-1], t[ilast]]: stored times themselves included (the <= of the walk)
            k = int(rng.integers(0, ilast))
            t = [T[k], np.nextafter(T[k], -np.inf), np.nextafter(T[k], np.inf), rng.uniform(T[0], T[ilast - 1])][int(rng.integers(0, 4))]
            if not (t - T[ilast - 1]) < 0.0:
                continue
            want = walk_left(T, t, ilast, T[ilast - 1], T[ilast])
            # the device enters the search after one step of the walk: t <= t[ilast-1]
            k, tlv, thv, probes = search_left(T, t, ilast - 1, T[ilast - 1])
            assert k == want[0] and thv == want[2]
            if k > 0:
                assert tlv == want[1]
            assert probes <= NP * (2 + 2 * int(np.ceil(np.log2(n))))


def test_right_search_ends_where_the_walk_ends():
    for T, rng in grids():
        n = len(T)
        for _ in range(400):
            ilast = int(rng.integers(1, n))
            k = int(rng.integers(ilast, n))
            t = [T[k], np.nextafter(T[k], -np.inf), np.nextafter(T[k], np.inf), T[-1] + 1.0][int(rng.integers(0, 4))]
            if not (t - T[ilast]) > 0.0:
                continue
            want = walk_right(T, t, ilast, T[ilast - 1], T[ilast])
            if ilast >= n - 1:
                continue
            # the device takes the first step of the walk itself, then searches
            indx, tprev, tcur = ilast + 1, T[ilast], T[ilast + 1]
            if indx < n - 1 and (t - tcur) > 0.0:
                indx, tprev, tcur, _ = search_right(T, t, indx, tcur)
            assert (indx, tprev, tcur) == want


def test_a_point_before_the_first_stored_time_ends_at_index_zero():
    T = np.r_[0.0, np.cumsum(np.full(99, 0.1))]
    k, tlv, thv, _ = search_left(T, -1.0, 57, T[57])
    assert k == 0 and thv == T[0] and walk_left(T, -1.0, 58, T[57], T[58])[0] == 0
