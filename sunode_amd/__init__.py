"""sunode_amd: MI355X-native batched BDF + adjoint engine behind the sunode API."""
from sunode_amd.symode import SympyProblem  # noqa: F401

__version__ = "0.1.0"
__all__ = ["SympyProblem"]
