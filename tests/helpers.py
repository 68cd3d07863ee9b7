"""Shared helpers for the test-suite (problem construction, oracle cache)."""
import functools

from tools.problem_cache import make_problem  # noqa: F401  (named problems, symbolic work cached on disk)


@functools.lru_cache(maxsize=None)
def make_oracle(name):
    from oracle.harness import Oracle
    # the 100-state callbacks are 3 MB of straight-line C: -O1 keeps the build at half a minute
    return Oracle(make_problem(name), tag=name, opt="-O1" if name == "network100" else "-O2")


def network_golden_points(golden_dir, name):
    """Points of tests/golden/callbacks_network.json (reference-generated: tools/make_golden_callbacks_network.py)
    with the rate matrix rebuilt from the repo-owned generator (the fixture stores its seed, not n^2 numbers)."""
    import json
    import os
    import numpy as np
    from tools.problems import std_normal
    with open(os.path.join(golden_dir, "callbacks_network.json")) as fh:
        g = json.load(fh)[name]
    n, pts = g["n"], g["points"]
    n_items = n * n + 4
    out = []
    for pt in pts:
        seed, stream, k = pt["K_seed"]
        K = np.abs(std_normal(seed, stream, len(pts) * n_items)).reshape(len(pts), n_items)[k, :n * n] / n
        out.append(dict(pt, K=K))
    return n, out


def check_matrix_summary(M, want, rtol=1e-13):
    """M (n x n, row = output) against the fixture's projections / diagonal / strided sample of the reference's matrix."""
    import numpy as np
    n = M.shape[0]
    u = 1.0 + 0.5 * np.cos(0.7 * np.arange(n))
    w = 1.0 + 0.5 * np.sin(1.3 * np.arange(n) + 0.2)
    for got, key in ((M @ u, "Mu"), (M.T @ w, "MTw"), (np.diag(M), "diag"), (M.ravel()[np.arange(0, n * n, 37)], "sample")):
        ref = np.array(want[key])
        np.testing.assert_allclose(got, ref, rtol=rtol, atol=64 * 2.3e-16 * np.abs(ref).max(), err_msg=key)
