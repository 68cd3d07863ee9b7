"""Pre-compile everything tests/test_shape_sweep.py needs: problems (symbolic work -> disk cache), default code
objects (adjoint + forward sensitivities) and oracle libraries.  Called by ``__graft_entry__.build()``; stand-alone:
``python tools/build_sweep.py [name ...]``."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _symbolic(name):
    from tools.problem_cache import make_problem
    t = time.time()
    make_problem(name)
    return name, time.time() - t


def _compile(name, kind):
    from oracle.harness import build_oracle_library
    from sunode_amd import _native
    from tools.problem_cache import make_problem
    src = make_problem(name).native_source()
    t = time.time()
    if kind == "oracle":
        out = build_oracle_library(src, name)
    elif kind == "sens":
        out = _native.build_code_object(src, sens=True)
    elif kind == "conservative":        # the differential guard's partner build (tests/test_guard.py runs the sweep's shapes through it)
        out = _native.build_code_object(src, compact=_native.default_compact_trajectory(src), safe=True)
    elif kind.startswith("small-batch"):     # AdjointSolver's mappings for small batches (_native.small_batch_group)
        out = _native.build_code_object(src, compact=True, group=kind.split(":")[1])
    else:
        out = _native.build_code_object(src, compact=_native.default_compact_trajectory(src))
    return "%s [%s]" % (name, kind), out, time.time() - t


def build(names=None, verbose=True):
    from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor
    from tools.sweep_cases import ADJOINT_CASES, SENS_CASES
    adj = [c[0] for c in ADJOINT_CASES if names is None or c[0] in names]
    sens = [c[0] for c in SENS_CASES if names is None or c[0] in names]
    every = sorted(set(adj) | set(sens), key=lambda s: (len(s), s))
    workers = max(2, min(8, (os.cpu_count() or 4) - 1))
    with ProcessPoolExecutor(max_workers=workers) as pool:          # sympy holds the GIL: processes
        for name, dt in pool.map(_symbolic, every[::-1]):
            if verbose and dt > 5:
                print("problem [%s]: %.0f s of sympy" % (name, dt))
    jobs = [(n, "oracle") for n in every] + [(n, "adjoint") for n in adj] + [(n, "sens") for n in sens] \
        + [(n, "conservative") for n in adj] + [(n, "small-batch:" + g) for n in adj if n == "rn5_8" for g in ("wave16", "wave8", "wave4")]
    failed = []
    with ThreadPoolExecutor(max_workers=workers) as pool:           # compiler subprocesses: threads
        futs = [(j, pool.submit(_compile, *j)) for j in jobs]
        for j, f in futs:
            try:
                label, out, dt = f.result()
                if verbose:
                    print("%s: %s (%.0f s)" % (label, os.path.basename(out), dt))
            except Exception as exc:        # noqa: BLE001 -- report every failing shape, then fail
                failed.append((j, exc))
                print("FAILED %s [%s]: %s" % (j[0], j[1], str(exc)[-1500:]))
    if failed:
        raise RuntimeError("shape sweep: %d build(s) failed: %s" % (len(failed), [j for j, _ in failed]))


if __name__ == "__main__":
    build(sys.argv[1:] or None)
