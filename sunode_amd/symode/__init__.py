from sunode_amd.symode.problem import SympyProblem  # noqa: F401

__all__ = ["SympyProblem"]
