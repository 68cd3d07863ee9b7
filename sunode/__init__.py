"""``sunode`` import surface over the MI355X engine (``sunode_amd``).

User code written against pymc-devs/sunode imports ``sunode.symode.SympyProblem``, ``sunode.solver.Solver`` /
``AdjointSolver`` / ``SolverError`` and ``sunode.wrappers.as_pytensor`` (reference ``sunode/__init__.py:3-7``,
``symode/__init__.py:1``, ``wrappers/__init__.py:1``).  This package makes those imports resolve to the modules
of ``sunode_amd`` -- the SAME module objects, no second copy -- so such code runs unchanged on the HIP engine.
Names of the reference that belong to its SUNDIALS binding layer (``empty_vector``, ``from_numpy``,
``empty_matrix``: cffi N_Vector / SUNMatrix wrappers) have no counterpart: the engine owns flat device buffers.
"""
import importlib
import importlib.abc
import importlib.util
import sys

import sunode_amd
from sunode_amd.symode import SympyProblem  # noqa: F401

__version__ = sunode_amd.__version__
__all__ = ["SympyProblem"]


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """``import sunode.x.y`` -> the module object of ``sunode_amd.x.y``."""

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith("sunode."):
            return None
        real = "sunode_amd." + fullname[len("sunode."):]
        try:
            found = importlib.util.find_spec(real)
        except (ImportError, ValueError):
            return None
        if found is None:
            return None
        return importlib.util.spec_from_loader(fullname, self, is_package=found.submodule_search_locations is not None)

    def create_module(self, spec):
        return importlib.import_module("sunode_amd." + spec.name[len("sunode."):])

    def exec_module(self, module):          # already executed under its real name
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

from sunode import solver, symode  # noqa: E402,F401  (reference sunode/__init__.py:5-6)
