"""Named test / bench problems as ``SympyProblem`` objects, with a disk cache of the symbolic work.

``make_problem(name)`` is what tests/helpers.py, ``__graft_entry__.build()`` and tools/code_object_budget.py share
(the build path must not import the test package).  Names: the keys of tools/problems.py ``PROBLEMS`` /
``EXTRA_PROBLEMS``, ``network100``, ``lv12``, ``chain<n>`` (bidiagonal decay chain) and the generated family ``rn<n>_<p>`` / ``rnb<n>_<p>`` (banded rate
matrix, |i - j| <= 2) of ``random_network``.

The derivation of a 128-state Jacobian takes sympy minutes; the pickled problem (cloudpickle: the right-hand sides are
closures) is kept under ``sunode_amd/_cache/problems/`` keyed by the hash of every source file that shapes it, so the
GPU box -- which receives ``_cache/`` with the snapshot -- does not repeat the work ``build()`` did.
"""
import functools
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(ROOT, "sunode_amd", "_cache", "problems")
_SOURCES = ["tools/problems.py", "sunode_amd/symode/problem.py", "sunode_amd/symode/codegen.py",
            "sunode_amd/symode/lambdify.py", "sunode_amd/csrc/sa_math.h", "sunode_amd/dtypesubset.py"]


def spec_of(name):
    from tools import problems as P
    if name == "network100":
        return P.network100()
    if name == "lv12":
        return P.LV12
    m = re.fullmatch(r"chain(\d+)", name)
    if m:
        return P.chain(int(m.group(1)))
    m = re.fullmatch(r"rn(b?)(\d+)_(\d+)", name)
    if m:
        return P.random_network(int(m.group(2)), int(m.group(3)), band=2 if m.group(1) else 0)
    return {**P.PROBLEMS, **P.EXTRA_PROBLEMS}[name]


@functools.lru_cache(maxsize=None)
def _source_key():
    h = hashlib.sha256()
    for rel in _SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    import sympy
    h.update(sympy.__version__.encode())
    return h.hexdigest()[:16]


@functools.lru_cache(maxsize=None)
def make_problem(name):
    from sunode_amd import SympyProblem
    path = os.path.join(_DIR, "%s_%s.pkl" % (name, _source_key()))
    if os.path.exists(path) and not os.environ.get("SA_NO_PROBLEM_CACHE"):
        try:
            import cloudpickle
            with open(path, "rb") as fh:
                prob = cloudpickle.load(fh)
            if isinstance(prob, SympyProblem):
                return prob
        except Exception:       # noqa: BLE001 -- a stale / truncated cache entry is rebuilt
            pass
    s = spec_of(name)
    prob = SympyProblem(s["params"], s["states"], s["rhs"], s["derivative_params"])
    prob.native_source()
    try:
        import cloudpickle
        os.makedirs(_DIR, exist_ok=True)
        tmp = "%s.tmp%d" % (path, os.getpid())
        with open(tmp, "wb") as fh:
            cloudpickle.dump(prob, fh)
        os.replace(tmp, path)
    except Exception:           # noqa: BLE001 -- the cache is an optimisation
        pass
    return prob
