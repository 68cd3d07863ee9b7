"""Shared helpers for the test-suite (problem construction, oracle cache)."""
import functools

from sunode_amd import SympyProblem
from tools.problems import EXTRA_PROBLEMS, PROBLEMS, network100


@functools.lru_cache(maxsize=None)
def make_problem(name):
    spec = network100() if name == "network100" else {**PROBLEMS, **EXTRA_PROBLEMS}[name]
    return SympyProblem(spec["params"], spec["states"], spec["rhs"], spec["derivative_params"])


@functools.lru_cache(maxsize=None)
def make_oracle(name):
    from oracle.harness import Oracle
    # the 100-state callbacks are 3 MB of straight-line C: -O1 keeps the build at half a minute
    return Oracle(make_problem(name), tag=name, opt="-O1" if name == "network100" else "-O2")
