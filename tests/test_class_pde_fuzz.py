"""Differential fuzz of the class PDEs through the REAL py-pde: random grids (1-3 axes, mixed periodicity, non-unit spacing),
random boundary conditions per face (Dirichlet / Neumann / mixed / curvature / position- and time-dependent expressions), random solver — hip (tests-only host shim: the
product's BC conversion, right-hand-side selection, stepper loops) against the reference's numpy backend on the same objects."""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

from refpath import REF  # noqa: E402
if not (REF / "pde").exists():
    pytest.skip("py-pde (reference) not available", allow_module_level=True)
if str(REF) not in sys.path:
    sys.path.append(str(REF))

import pde  # noqa: E402
import shimlib  # noqa: E402
from helpers import max_rel  # noqa: E402


def _random_face(rng, axes=""):
    kind = rng.integers(6 if axes else 4)
    if kind == 4:   # expression conditions: position and time dependent
        return {"value_expression": f"0.1 * sin(3 * t) + 0.05 * {axes[rng.integers(len(axes))]}"}
    if kind == 5:
        return {"derivative_expression": f"0.1 * cos(t) * {axes[rng.integers(len(axes))]}"}
    if kind == 0:
        return {"value": float(rng.uniform(-0.5, 0.5))}
    if kind == 1:
        return {"derivative": float(rng.uniform(-0.3, 0.3))}
    if kind == 2:
        return {"type": "mixed", "value": float(rng.uniform(0.1, 1.0)), "const": float(rng.uniform(-0.3, 0.3))}
    return {"curvature": float(rng.uniform(-0.2, 0.2))}


def _random_bc(rng, grid):
    bc = {}
    for ax, per in zip(grid.axes, grid.periodic):
        if per:
            bc[ax] = "periodic"
        else:
            others = "".join(a for a in grid.axes if a != ax)
            bc[f"{ax}-"], bc[f"{ax}+"] = _random_face(rng, others), _random_face(rng, others)
    return bc


@pytest.mark.parametrize("seed", range(60))
def test_random_class_pde_runs_match_the_reference(seed, monkeypatch):
    monkeypatch.setitem(pde.config, "default_backend", "scipy")   # operators of the numpy path (numba is not installed here)
    rng = np.random.default_rng(3000 + seed)
    nd = 1 + seed % 3
    shape = [int(rng.integers(5, 14)) for _ in range(nd)]
    periodic = [bool(rng.integers(2)) for _ in range(nd)]
    dx = float(rng.choice([0.5, 1.0, 2.0]))                        # (the scipy operators want one spacing for all axes)
    grid = pde.CartesianGrid([[0, dx * n] for n in shape], shape, periodic=periodic)
    state = pde.ScalarField.random_uniform(grid, -0.5, 0.5, rng=rng)
    which = seed % 4
    if which == 0:
        eq = pde.DiffusionPDE(diffusivity=float(rng.uniform(0.2, 1.5)), bc=_random_bc(rng, grid))
    elif which == 1:
        eq = pde.CahnHilliardPDE(interface_width=float(rng.uniform(0.5, 1.5)), bc_c=_random_bc(rng, grid), bc_mu=_random_bc(rng, grid))
    elif which == 2:
        eq = pde.AllenCahnPDE(interface_width=float(rng.uniform(0.5, 1.5)), mobility=float(rng.uniform(0.5, 1.5)), bc=_random_bc(rng, grid))
    else:
        eq = pde.SwiftHohenbergPDE(rate=0.1, kc2=float(rng.uniform(0.2, 1.0)), delta=float(rng.uniform(0, 1)), bc=_random_bc(rng, grid),
                                   bc_lap=_random_bc(rng, grid))
    solver = ["euler", "runge-kutta"][int(rng.integers(2))]
    adaptive = (bool(seed % 5 == 0) and solver == "runge-kutta") or (bool(seed % 3 == 1) and solver == "euler")   # RKF45 / the reference's adaptive Euler
    dt = 1e-3 * dx**4
    kw = dict(t_range=12 * dt, dt=dt, solver=solver, tracker=None, ret_info=True)
    if adaptive:
        kw["adaptive"] = True
    ref, iref = eq.solve(state, backend="numpy", **kw)
    with shimlib.use_shim(fused=bool(seed % 2)):
        import pde_hip.pypde_plugin  # noqa: F401

        out, info = eq.solve(state, backend="hip", **kw)
        got = np.array(out.data)   # (the result is linked to its device copy: read it while that library is in place)
    assert info["solver"]["steps"] == iref["solver"]["steps"]
    assert np.isfinite(ref.data).all()
    assert max_rel(got, ref.data) < 1e-9, (type(eq).__name__, shape, periodic, solver)


def _random_nonlinear_face(rng, axes):
    """Conditions that are NOT affine in the adjacent value (they read the field when they are applied, pde_hip/bc_expr.py)."""
    a, b = float(rng.uniform(0.1, 0.5)), float(rng.uniform(-0.2, 0.2))
    other = f" + {b:.3f} * {axes[rng.integers(len(axes))]}" if axes else ""
    return [
        {"derivative_expression": f"-{a:.3f} * value**3{other}"},
        {"value_expression": f"{a:.3f} * tanh(value) + {b:.3f} * sin(2 * t)"},
        {"virtual_point": f"value / (1 + {a:.3f} * value**2){other}"},
        {"type": "mixed_expression", "value": f"0.5 + {a:.3f} * value**2", "const": f"{b:.3f}"},
    ][int(rng.integers(4))]


@pytest.mark.parametrize("seed", range(24))
def test_random_runs_with_conditions_that_read_the_field(seed, monkeypatch):
    """Diffusion / Allen-Cahn / Cahn-Hilliard (conditions of c) with random faces that depend non-linearly on the adjacent value,
    mixed with the affine kinds above: every right-hand side - the Runge-Kutta stage inputs included - sees the conditions of ITS
    input field.  hip (host shim) against the reference's numpy backend: equal step counts, <= 1e-9."""
    monkeypatch.setitem(pde.config, "default_backend", "scipy")
    rng = np.random.default_rng(5000 + seed)
    nd = 1 + seed % 3
    shape = [int(rng.integers(5, 12)) for _ in range(nd)]
    periodic = [False] + [bool(rng.integers(2)) for _ in range(nd - 1)]
    dx = float(rng.choice([0.5, 1.0, 2.0]))
    grid = pde.CartesianGrid([[0, dx * n] for n in shape], shape, periodic=periodic)
    state = pde.ScalarField.random_uniform(grid, -0.5, 0.5, rng=rng)

    def bc_nonlinear():
        bc = {}
        for ax, per in zip(grid.axes, grid.periodic):
            if per:
                bc[ax] = "periodic"
                continue
            others = "".join(a for a in grid.axes if a != ax)
            for side in "-+":
                bc[ax + side] = _random_nonlinear_face(rng, others) if rng.random() < 0.6 else _random_face(rng, others)
        return bc

    which = seed % 3
    if which == 0:
        eq = pde.DiffusionPDE(diffusivity=float(rng.uniform(0.2, 1.5)), bc=bc_nonlinear())
    elif which == 1:
        eq = pde.AllenCahnPDE(interface_width=float(rng.uniform(0.5, 1.5)), mobility=float(rng.uniform(0.5, 1.5)), bc=bc_nonlinear())
    else:
        eq = pde.CahnHilliardPDE(interface_width=float(rng.uniform(0.5, 1.5)), bc_c=bc_nonlinear(), bc_mu=_random_bc(rng, grid))
    solver = ["euler", "runge-kutta"][int(rng.integers(2))]
    adaptive = (bool(seed % 4 == 0) and solver == "runge-kutta") or (bool(seed % 3 == 1) and solver == "euler")
    dt = 1e-3 * dx**4
    kw = dict(t_range=6 * dt, dt=dt, solver=solver, tracker=None, ret_info=True)
    if adaptive:
        kw["adaptive"] = True
    ref, iref = eq.solve(state, backend="numpy", **kw)
    with shimlib.use_shim(fused=bool(seed % 2)):
        import pde_hip.pypde_plugin  # noqa: F401

        out, info = eq.solve(state, backend="hip", **kw)
        got = np.array(out.data)
    assert info["solver"]["steps"] == iref["solver"]["steps"]
    assert np.isfinite(ref.data).all()
    assert max_rel(got, ref.data) < 1e-9, (type(eq).__name__, shape, periodic, solver)
