"""Minimal stand-in for pytensor (absent in this image) so that the *numeric* halves of the Ops in
sunode_amd.wrappers.as_pytensor -- their ``perform`` methods -- can be exercised on the GPU.  It only provides
the names the wrapper module touches at import and in ``perform``; graph construction (``solve_ivp``, ``grad``)
needs the real package.  Never on sys.path outside tests/test_pytensor_ops.py."""
