"""Minimal stand-in for pytensor (absent in this image): a tiny lazy graph -- Variables, Apply nodes, Ops with
``perform`` -- with just the tensor vocabulary sunode_amd.wrappers.as_pytensor uses, plus ``evaluate`` to run a
graph numerically.  It lets the tests execute ``solve_ivp`` graph construction and the ``grad`` wiring of the Ops
(forward Op -> backward Op -> EvalRhs) end to end; with the real pytensor installed the tests use that instead.
Never on sys.path outside tests/test_pytensor_ops.py."""
from pytensor.graph.basic import evaluate  # noqa: F401
