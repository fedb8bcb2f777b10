"""The hip backend as a plugin of the REAL py-pde (CPU checks; skipped where py-pde is absent).

In the build container py-pde lives at /root/reference (read-only); on the GPU box it does not
exist, so these tests only cover what can be verified without a device: registration, config
linking, class attributes, operator registry, conversion of real py-pde BoundariesList objects into
the C face table (checked against py-pde's own numpy ghost cells through the oracle), RHS
recognition, and the loud failure without a GPU.
"""

from __future__ import annotations

import sys
from pathlib import Path

import ctypes as C

import numpy as np
import pytest

from refpath import REF  # noqa: E402
if not (REF / "pde").exists():
    pytest.skip("py-pde (reference) not available", allow_module_level=True)
if str(REF) not in sys.path:
    sys.path.append(str(REF))

import pde  # noqa: E402
from helpers import HostBuf, face_mask  # noqa: E402

import pde_hip.pypde_plugin as plugin  # noqa: E402
from oracle import pde_oracle as O  # noqa: E402
from pde_hip import _abi, _lib  # noqa: E402
from pde_hip.backend import _match_expression_rhs, convert_bcs  # noqa: E402


def test_registration_and_class_attributes():
    from pde.backends import backend_registry

    assert "hip" in backend_registry._packages and backend_registry._classes["hip"] is plugin.HipBackend
    cls = plugin.HipBackend
    assert issubclass(cls, pde.backends.base.BackendBase)
    assert cls.implementation == "hip" and cls.copy_data is True and cls.supports_mpi is False
    assert pde.config["backend"]["hip"]["device"] == -1 and pde.config["backend"]["hip"]["resident_state"] is True   # -1: process default (LOCAL_RANK or 0)
    assert "hip_slab" in pde.solvers.registered_solvers()
    ops = cls._operators[pde.CartesianGrid]
    assert {"laplace", "gradient", "divergence", "gradient_squared", "vector_laplace"} <= set(ops)
    plugin.register()  # idempotent


def test_backend_is_lazy_and_compute_fails_loudly_without_gpu():
    """Construction must not need a device (grid.operators instantiates every backend); compute must (no fallback)."""
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    backend = pde.backends.backend_registry.get_backend("hip")
    assert "laplace" in pde.UnitGrid([4, 4]).operators
    with pytest.raises(RuntimeError, match="no HIP device"):
        backend.device_name
    with pytest.raises(RuntimeError, match="no HIP device"):
        pde.ScalarField(pde.UnitGrid([4, 4]), 1.0).laplace("auto_periodic_neumann", backend="hip")
    # ... and py-pde's other backends are unaffected
    assert pde.backends.backend_registry.get_backend("numpy").name == "numpy"


@pytest.mark.parametrize("bc", [
    "auto_periodic_neumann",
    {"x-": {"value": 1.5}, "x+": {"derivative": 0.3}, "y": "periodic"},
    {"x-": {"value": "sin(y)"}, "x+": {"type": "mixed", "value": 2.0, "const": 0.5}, "y": "periodic"},
    {"x": "extrapolate", "y": "periodic"},
])
def test_real_boundaries_to_face_table(bc):
    """convert_bcs on py-pde's own BoundariesList + oracle ghost setter == py-pde's numpy set_ghost_cells."""
    grid = pde.CartesianGrid([[0, 3], [1, 4.5]], [6, 5], periodic=[False, True])
    field = pde.ScalarField.random_uniform(grid, rng=np.random.default_rng(0))
    bcs = grid.get_boundary_conditions(bc, rank=0)
    table = convert_bcs(bcs, upload=HostBuf)
    full = field._data_full.copy()
    full[0, :] = full[-1, :] = full[:, 0] = full[:, -1] = 0
    O.set_ghost_cells(_abi.make_grid(grid.shape, grid.discretization, np.float64), 1, table.c, full)
    field.set_ghost_cells(bc)
    mask = face_mask(grid)
    np.testing.assert_array_equal(full[mask], field._data_full[mask])


def test_unsupported_boundaries_raise_not_implemented():
    grid = pde.UnitGrid([4, 4])
    bcs = grid.get_boundary_conditions({"value_expression": "t"}, rank=0)
    with pytest.raises(NotImplementedError, match="does not support boundary condition"):
        convert_bcs(bcs, upload=HostBuf)       # the plain converter knows constant conditions only ...
    from pde_hip.bc_expr import convert_bcs_with_expressions

    table = convert_bcs_with_expressions(bcs, upload=HostBuf)   # ... expressions affine in `value` become coefficient arrays
    assert table.time_dependent
    # ... expressions that are not: arrays with A = F(value), B = 0 that are rewritten from the field they are applied to
    nonlin = convert_bcs_with_expressions(grid.get_boundary_conditions({"virtual_point": "value**2"}, rank=0), upload=HostBuf)
    assert nonlin.time_dependent and nonlin.reads_value and not table.reads_value
    with pytest.raises(RuntimeError, match="need the field"):
        nonlin.update({"t": 0.0})
    full = np.arange(36.0).reshape(6, 6)
    nonlin.update({"t": 0.0}, state=full)
    np.testing.assert_array_equal(nonlin.keepalive[-2].arr, full[1:-1, -2] ** 2)    # face y+: the adjacent cells are column -2

    # conditions given as Python functions are probed on the host (before every right-hand side): affine ones work ...
    def user_bc(value, dx, x, y, t):
        return 0.5 * value + np.sign(x) * t

    table = convert_bcs_with_expressions(grid.get_boundary_conditions({"virtual_point": user_bc}, rank=0), upload=HostBuf)
    assert table.time_dependent

    def user_value(value, dx, x, y, t):
        return np.sin(y)

    ref = convert_bcs_with_expressions(grid.get_boundary_conditions({"value_expression": "sin(y)"}, rank=0), upload=HostBuf)
    fun = convert_bcs_with_expressions(grid.get_boundary_conditions({"type": "value_expression", "value": user_value}, rank=0), upload=HostBuf)
    fun.update({"t": 0.3})
    for k in range(4):   # exactly the coefficient arrays of the expression form (2 f, -1)
        n = grid.shape[1 - k // 2]
        for name in ("const_arr", "factor1_arr"):
            a = np.ctypeslib.as_array((C.c_double * n).from_address(getattr(ref.c[k], name)))
            b = np.ctypeslib.as_array((C.c_double * n).from_address(getattr(fun.c[k], name)))
            np.testing.assert_array_equal(a, b)

    # ... functions that are not affine in the adjacent value do not
    with pytest.raises(NotImplementedError, match="not affine"):
        convert_bcs_with_expressions(grid.get_boundary_conditions({"virtual_point": lambda value, dx, x, y, t: value**2}, rank=0), upload=HostBuf)


@pytest.mark.parametrize("bc", [
    {"x-": {"value_expression": "sin(y) + t"}, "x+": {"derivative_expression": "0.1 * y * t"}, "y": "periodic"},
    {"x": {"type": "mixed_expression", "value": "1 + y", "const": "cos(t)"}, "y": "periodic"},
    {"x-": {"virtual_point": "2 * value - y"}, "x+": {"type": "virtual_point", "value": "value + x", "value_cell": 1}, "y": "periodic"},
])
def test_expression_boundaries_to_face_table(bc):
    """Expression conditions lowered to coefficient arrays + oracle ghost setter == py-pde's numpy set_ghost_cells."""
    from pde_hip.bc_expr import convert_bcs_with_expressions

    grid = pde.CartesianGrid([[0, 3], [1, 4.5]], [6, 5], periodic=[False, True])
    field = pde.ScalarField.random_uniform(grid, rng=np.random.default_rng(0))
    table = convert_bcs_with_expressions(grid.get_boundary_conditions(bc, rank=0), upload=HostBuf)
    for t in (0.0, 0.4):
        full = field._data_full.copy()
        full[0, :] = full[-1, :] = full[:, 0] = full[:, -1] = 0
        table.update({"t": t})
        O.set_ghost_cells(_abi.make_grid(grid.shape, grid.discretization, np.float64), 1, table.c, full)
        field.set_ghost_cells(bc, args={"t": t})
        mask = face_mask(grid)
        np.testing.assert_allclose(full[mask], field._data_full[mask], rtol=1e-14, atol=1e-14)


def test_rhs_recognition_on_real_pde_objects():
    """The attributes make_rhs_spec reads exist on the real classes; expression PDE is matched."""
    eq = pde.DiffusionPDE(0.7, bc="auto_periodic_neumann")
    assert (eq.__class__.__name__, eq.diffusivity, eq.bc) == ("DiffusionPDE", 0.7, "auto_periodic_neumann")
    from pde_hip.backend import pde_bc_for, pde_expression

    gen2 = pde.PDE({"c": "∇²c"}, bc={"value": 1.0}, bc_ops={"c:laplace": {"value": 2.0}, "gradient": "periodic"})
    assert not hasattr(gen2, "bc") and not hasattr(gen2, "bc_ops")       # only `bcs` exists
    assert pde_bc_for(gen2, "c", "laplace") == {"value": 2.0} and pde_bc_for(gen2, "c", "gradient") == "periodic"
    assert pde_bc_for(gen2, "c", "gradient_squared") == {"value": 1.0} and pde_expression(gen2, "c") == "laplace(c)"
    ch = pde.CahnHilliardPDE(interface_width=0.5)
    assert ch.__class__.__name__ == "CahnHilliardPDE" and ch.interface_width == 0.5 and hasattr(ch, "bc_c") and hasattr(ch, "bc_mu")
    gen = pde.PDE({"c": "laplace(c**3 - c - laplace(c))"})
    (var, expr), = gen.rhs.items()
    assert _match_expression_rhs(str(expr), var, dict(gen.consts)) == (_abi.RHS_CAHN_HILLIARD, 1.0)
    solver = pde.solvers.EulerSolver(eq, backend="numpy")
    assert solver.__class__.__name__ == "EulerSolver" and hasattr(solver, "adaptive") and hasattr(solver, "tolerance")
    rk = pde.solvers.RungeKuttaSolver(eq, backend="numpy", adaptive=True)
    assert (rk.dt_min, rk.dt_max, rk.tolerance) == (1e-10, 1e10, 1e-4)
