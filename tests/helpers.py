"""Shared helpers for the test-suite (problem construction, oracle cache)."""
import functools

from sunode_amd import SympyProblem
from tools.problems import PROBLEMS


@functools.lru_cache(maxsize=None)
def make_problem(name):
    spec = PROBLEMS[name]
    return SympyProblem(spec["params"], spec["states"], spec["rhs"], spec["derivative_params"])


@functools.lru_cache(maxsize=None)
def make_oracle(name):
    from oracle.harness import Oracle
    return Oracle(make_problem(name), tag=name)
