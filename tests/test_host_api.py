"""Host-side logic that runs without a GPU: C-ABI library + symbols, solver construction,
argument validation, option handling, code-object build (cross-compiled for gfx950)."""
import ctypes
import os
import pickle
import re

import numpy as np
import pytest

from tests.helpers import make_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_library_exports_every_declared_symbol():
    """libsunode_amd.so loads on a CPU-only host and exports what include/sunode_amd.h declares."""
    from sunode_amd import _native
    L = _native.load_library()
    header = open(os.path.join(ROOT, "include", "sunode_amd.h")).read()
    declared = set(re.findall(r"\b(sa_[a-z_]+)\s*\(", header))
    declared -= {"sa_solver", "sa_options"}
    assert declared == set(_native.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    want = int(re.search(r"#define SA_ABI_VERSION (\d+)", header).group(1))
    assert L.sa_abi_version() == want == 3         # 3: arena layout by kernel family (sa_traj_point_major)
    abi = open(os.path.join(ROOT, "sunode_amd", "csrc", "sa_device_abi.h")).read()
    assert int(re.search(r"#define SA_DEVICE_ABI_VERSION (\d+)", abi).group(1)) == want


def test_native_create_fails_loudly_without_gpu_or_code_object():
    from sunode_amd import _native
    L = _native.load_library()
    opt = _native._Options()
    opt.struct_size = ctypes.sizeof(_native._Options)
    atol = (ctypes.c_double * 2)(1e-8, 1e-8)
    opt.atol = ctypes.cast(atol, ctypes.POINTER(ctypes.c_double))
    opt.traj_capacity = 16
    h = ctypes.c_void_p()
    rc = L.sa_solver_create(b"/nonexistent/problem.hsaco", ctypes.byref(opt), ctypes.byref(h))
    assert rc < 0 and not h
    assert len(L.sa_last_error()) > 0


def test_code_object_is_gfx950_and_has_no_scratch():
    """The per-lane integrator state must stay in registers (see bdf_kernels.hip on SROA)."""
    import subprocess
    from sunode_amd import _native
    co = _native.build_code_object(make_problem("lv").native_source())
    notes = subprocess.run([os.path.join(_native.LLVM_BIN, "llvm-readelf"), "--notes", co],
                           capture_output=True, text=True, check=True).stdout
    assert "gfx950" in notes
    kernels = re.findall(r"\.name:\s+(sa_k_\w+)", notes)
    assert {"sa_k_forward", "sa_k_backward", "sa_k_eval", "sa_k_math"} <= set(kernels)
    scratch = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)]
    # the integrator state is > 1 KiB per lane; a few bytes of compiler temporaries are tolerated
    assert scratch and max(scratch) <= 64


def test_solver_construction_like_reference_tests():
    """sunode/test_solve.py:7-78: scalar, empty and nested params/states construct."""
    from sunode_amd import SympyProblem
    from sunode_amd.solver import AdjointSolver, Solver
    Solver(SympyProblem({"b": ()}, {"x": ()}, lambda t, y, p: {"x": y.x}, derivative_params=[]))
    Solver(SympyProblem({}, {"x": ()}, lambda t, y, p: {"x": y.x}, derivative_params=[]))
    Solver(SympyProblem({"a": {"b": ()}}, {"x": ()}, lambda t, y, p: {"x": y.x + p.a.b}, derivative_params=[]))
    prob = SympyProblem({"a": {"b": ()}}, {"x": {"y": {"z": ()}}},
                        lambda t, y, p: {"x": {"y": {"z": y.x.y.z + p.a.b}}}, derivative_params=[("a", "b")])
    s = AdjointSolver(prob)
    y, g, lam = s.make_output_buffers(np.linspace(0, 1))
    assert y.shape == (50, 1) and g.shape == (1,) and lam.shape == (1,)
    s.set_params_dict({"a": {"b": 0.2}})
    assert s.get_params_dict()["a"]["b"] == 0.2
    assert s.derivative_params_dtype.names == ("a",) and s.remainder_params_dtype.itemsize == 0


def test_unsupported_reference_options_raise():
    from sunode_amd.solver import AdjointSolver, Solver
    prob = make_problem("lv")
    with pytest.raises(ValueError):
        Solver(prob, sens_mode="staggered1")      # as the reference (solver.py:365-366)
    with pytest.raises(ValueError):
        Solver(prob, sens_mode="bogus")
    with pytest.raises(NotImplementedError):
        Solver(prob, solver="ADAMS")
    with pytest.warns(RuntimeWarning):
        Solver(prob, linear_solver="spgmr")      # accepted like the reference, served by the dense analytic path
    with pytest.raises(ValueError):
        Solver(prob, linear_solver="nope")
    with pytest.raises(ValueError):
        AdjointSolver(prob, interpolation="spline")
    with pytest.raises(ValueError):
        AdjointSolver(prob, adjoint_solver="RK4")
    with pytest.raises(ValueError):
        Solver(prob, abstol=np.ones(3))          # wrong length for a 2-state problem
    with pytest.raises(ValueError):
        Solver(prob, constraints=[3.0, 0.0])     # CVodeSetConstraints accepts 0, +-1, +-2
    with pytest.raises(ValueError):
        Solver(prob, constraints=[1.0, 1.0], sens_mode="simultaneous")


def test_parameter_plumbing_matches_reference_semantics():
    from sunode_amd.solver import Solver
    prob = make_problem("seir")
    s = Solver(prob)
    full = np.zeros((), dtype=prob.params_dtype)
    full["beta"] = [1, 2, 3, 4]
    full["C"] = np.arange(16.0).reshape(4, 4)
    full["rates"]["sigma"] = 5
    full["rates"]["nu"] = 8
    s.set_params(full)
    ps, pr = prob.flat_params(s._user_data)
    assert ps.tolist() == [1, 2, 3, 4, 5, 0, 0, 8] and pr.tolist() == list(np.arange(16.0))
    sub = np.zeros((), dtype=s.derivative_params_dtype)
    sub["beta"] = 9
    sub["rates"]["gamma"] = 7
    s.set_derivative_params(sub)
    ps, pr = prob.flat_params(s._user_data)
    assert ps.tolist() == [9, 9, 9, 9, 0, 7, 0, 0] and pr.tolist() == list(np.arange(16.0))
    rem = np.zeros((), dtype=s.remainder_params_dtype)
    rem["C"] = 1.5
    s.set_remaining_params(rem)
    assert (prob.flat_params(s._user_data)[1] == 1.5).all()


def test_solver_pickles_like_the_reference():
    from sunode_amd.solver import Solver
    s = Solver(make_problem("lv"), abstol=1e-7, reltol=1e-6)
    s.set_params_dict({"alpha": 0.1, "beta": 0.2, "gamma": 0.3, "delta": 0.4})
    # SympyProblem holds the user's rhs function; module-level functions pickle fine
    s2 = pickle.loads(pickle.dumps(s))
    assert float(s2._rtol) == 1e-6 and s2.get_params_dict()["delta"] == 0.4


def test_pytensor_wrapper_is_import_guarded():
    """pytensor is optional (absent in this image): the wrapper must fail with a clear ImportError,
    and importing the package itself must not pull it in."""
    import importlib
    import sunode_amd  # noqa: F401
    try:
        import pytensor  # noqa: F401
        have = True
    except ImportError:
        have = False
    if have:
        mod = importlib.import_module("sunode_amd.wrappers.as_pytensor")
        assert hasattr(mod, "solve_ivp") and hasattr(mod, "SolveODEAdjoint")
    else:
        with pytest.raises(ImportError, match="pytensor"):
            importlib.import_module("sunode_amd.wrappers.as_pytensor")


def test_sunode_import_surface_is_the_engine():
    """`import sunode...` as user code written for pymc-devs/sunode does (reference sunode/__init__.py:3-7):
    the names resolve to the sunode_amd modules themselves."""
    import sunode
    import sunode.solver
    import sunode.symode
    import sunode.symode.problem
    import sunode_amd.solver
    from sunode.symode import SympyProblem as P1
    from sunode_amd import SympyProblem as P2
    assert P1 is P2 and sunode.SympyProblem is P2
    assert sunode.solver.Solver is sunode_amd.solver.Solver
    assert sunode.solver.AdjointSolver is sunode_amd.solver.AdjointSolver
    assert issubclass(sunode.solver.SolverError, RuntimeError)
    assert sunode.symode.problem.SympyProblem is P2
    import sunode.wrappers                      # pytensor itself stays optional
    assert sunode.wrappers.__name__ == "sunode_amd.wrappers"


def test_solution_variables_follow_the_leaf_table():
    """Content of the reference's solution_to_xarray (problem.py:100-145) without xarray: names, dims, values."""
    from tests.helpers import make_problem
    prob = make_problem("seir")
    ud = prob.make_user_data()
    vals = np.arange(24.0)
    prob.update_params(ud, vals.view(prob.params_dtype)[0])
    sol = np.arange(3 * 16.0).reshape(3, 16)
    v = prob.solution_variables([0.0, 1.0, 2.0], sol, ud)
    assert list(v)[:5] == ["time", "solution_S", "solution_E", "solution_I", "solution_R"]
    assert v["solution_I"][0] == ("time", "_I_dim0__") and np.array_equal(v["solution_I"][1], sol[:, 8:12])
    assert v["parameters_C"][0] == ("_C_dim0__", "_C_dim1__")
    assert np.array_equal(v["parameters_C"][1], vals[4:20].reshape(4, 4))
    assert v["parameters_rates_gamma"][0] == () and float(v["parameters_rates_gamma"][1]) == 21.0
    packed = prob.solution_variables([0.0, 1.0, 2.0], sol, ud, unstack_state=False, unstack_params=False)
    assert packed["solution"][1].dtype == prob.state_dtype and packed["solution"][1].shape == (3,)
    assert packed["parameters"][1].dtype == prob.params_dtype


def test_hermite_kernel_selection_with_many_parameters():
    """Few states but more differentiated parameters than one lane carries: the build (Hermite or not) must
    fall through to the lane-group kernel instead of failing (kernel_variant)."""
    from sunode_amd import _native
    src = "#define SA_N_STATES 3\n#define SA_N_SUB 10\n#define SA_N_REM 0\n"
    assert _native.kernel_variant(src, hermite=True) == ("bdf_wave.hip", 4)
    assert _native.kernel_variant(src, hermite=False)[0] == "bdf_wave.hip"
    small = "#define SA_N_STATES 3\n#define SA_N_SUB 3\n#define SA_N_REM 0\n"
    assert _native.kernel_variant(small, hermite=True) == ("bdf_kernels.hip", 1)     # register kernel carries Hermite
    assert _native.kernel_variant(small) == ("bdf_kernels.hip", 1)


def test_default_arena_record_format_and_build_flags(monkeypatch):
    """AdjointSolver's default record format (compact {order, t, y[n]} from three states on in the register-resident
    kernels, table records for n = 2, Hermite data and the memory-resident kernel) and the per-build code-generation
    flags (the spill-splitting allocator mode only in adjoint builds; the lane-group kernels without the ILP scheduler)."""
    from sunode_amd import _native
    hdr = "#define SA_N_STATES %d\n#define SA_N_SUB %d\n#define SA_N_REM 0\n"
    monkeypatch.delenv("SA_FORCE_GROUP", raising=False)
    assert _native.default_compact_trajectory(hdr % (2, 2)) is False                 # Lotka-Volterra: table records
    assert _native.default_compact_trajectory(hdr % (3, 3)) is True                  # Robertson
    assert _native.default_compact_trajectory(hdr % (16, 8)) is True                 # SEIR: lean lane groups
    assert _native.default_compact_trajectory(hdr % (100, 4)) is True                # network100: workgroup per instance
    assert _native.default_compact_trajectory(hdr % (3, 3), hermite=True) is False
    assert _native.default_compact_trajectory(hdr % (200, 4)) is False               # memory-resident kernel
    assert "split-spill-mode" in _native.ADJOINT_CODEGEN_FLAGS
    assert "split-spill-mode" not in _native.DEFAULT_CODEGEN_FLAGS and "iterative-ilp" in _native.DEFAULT_CODEGEN_FLAGS
    assert "sched-strategy" not in _native.WAVE_CODEGEN_FLAGS
    assert "misched-cluster" in _native.SMALL_GROUP_CODEGEN_FLAGS and "misched-cluster" not in _native.WAVE_CODEGEN_FLAGS
    # the cache key separates the record formats and the sensitivity builds of one problem
    src = hdr % (3, 3)
    keys = {_native.code_object_path(src), _native.code_object_path(src, compact=True),
            _native.code_object_path(src, sens=True), _native.code_object_path(src, hermite=True)}
    assert len(keys) == 4
