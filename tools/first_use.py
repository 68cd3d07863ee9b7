"""First-use cost of a model: what a user pays before the first batch (missing #9).

The reference JIT-compiles its numba callbacks at ``Solver`` construction (/root/reference/sunode/solver.py:242-317,
symode/problem.py:251-433); here a new model costs (1) the symbolic derivation + C generation (``SympyProblem`` +
``native_source()``), (2) the default gfx950 code object and (3) the conservative partner build of the differential
guard (``NativeSolver`` compiles (2) and (3) side by side).  All three are cached on disk afterwards.

    python tools/first_use.py [lv robertson seir network100]      -> profiles/r06_first_use.json

Measures from a cold cache (``force=True`` rebuilds; the symbolic step bypasses tools/problem_cache.py).  hipcc
cross-compiles: no GPU needed, the numbers are host-CPU seconds of THIS machine (core count recorded).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(name):
    from concurrent.futures import ThreadPoolExecutor
    from sunode_amd import SympyProblem, _native
    from tools.problem_cache import spec_of
    s = spec_of(name)
    t0 = time.perf_counter()
    prob = SympyProblem(s["params"], s["states"], s["rhs"], s["derivative_params"])
    t1 = time.perf_counter()
    src = prob.native_source()
    t2 = time.perf_counter()
    kw = dict(compact=_native.default_compact_trajectory(src))
    _native.build_code_object(src, force=True, **kw)
    t3 = time.perf_counter()
    _native.build_code_object(src, force=True, safe=True, **kw)
    t4 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=2) as pool:          # what NativeSolver does on a cold cache
        list(pool.map(lambda safe: _native.build_code_object(src, force=True, safe=safe, **kw), (False, True)))
    t5 = time.perf_counter()
    fname, group = _native.kernel_variant(src)
    return {"n_states": prob.n_states, "n_params": prob.n_params, "kernel": "%s, %d lane(s) per instance" % (fname, group),
            "generated_source_kb": round(len(src) / 1024.0, 1),
            "sympy_derivation_s": round(t1 - t0, 2), "code_generation_s": round(t2 - t1, 2),
            "default_build_s": round(t3 - t2, 2), "conservative_build_s": round(t4 - t3, 2),
            "both_builds_side_by_side_s": round(t5 - t4, 2),
            "first_batch_ready_after_s": round((t2 - t0) + (t5 - t4), 2)}


def main():
    from sunode_amd import _native
    names = sys.argv[1:] or ["lv", "robertson", "seir", "network100"]
    out = {"host_cores": os.cpu_count(), "toolchain": _native.toolchain_id()["hash"],
           "what": "cold-cache seconds on this host: SympyProblem() + native_source(), then the default and the "
                   "conservative (differential guard) code object; cached on disk afterwards (second use: < 0.1 s)",
           "configs": {}}
    for name in names:
        out["configs"][name] = measure(name)
        print(name, out["configs"][name], flush=True)
    path = os.path.join(ROOT, "profiles", "r06_first_use.json")
    if os.path.exists(path) and sys.argv[1:]:                # a partial run updates the record
        with open(path) as fh:
            old = json.load(fh)
        old["configs"].update(out["configs"])
        out["configs"] = old["configs"]
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
