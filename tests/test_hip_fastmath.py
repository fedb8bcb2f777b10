"""GPU tests of the opt-in arithmetic mode (`pdehip_set_fastmath`, ``backend.fastmath``, ``config["backend.hip.fastmath"]``): the stencil kernels
compiled with FMA contraction, like the reference's numba backend under its default ``fastmath`` (``pde/backends/numba/utils.py:330-336``).

The default mode is bit-exact against the oracle (every other GPU test); here: the contracted build agrees with the exact one within north_star's
tolerance (1e-10 relative to the field's scale) after 200 Euler steps and on the other schemes, it really is a different build (the bits differ
somewhere), and switching back restores bit-exactness.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

import pde_hip

pytestmark = pytest.mark.gpu

TOL = 1e-10


def _solve(backend, eq, state, **kw):
    return eq.solve(state, backend=backend, **kw).data


@pytest.fixture
def backend():
    b = pde_hip.get_backend("hip")
    yield b
    b.fastmath = None
    _ = b._lib   # applies the default mode again


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.mark.parametrize("shape,periodic", [((96, 96, 128), True), ((64, 72, 130), [True, False, False]), ((256, 384), True), ((100, 100, 100), False)])
def test_contracted_build_agrees_with_the_exact_one_after_200_steps(backend, shape, periodic):
    grid = pde_hip.UnitGrid(shape, periodic=periodic)
    state = pde_hip.ScalarField.random_uniform(grid, -1, 1, rng=np.random.default_rng(3))
    eq = pde_hip.DiffusionPDE(0.9)
    backend.fastmath = False
    exact = _solve(backend, eq, state, t_range=20.0, dt=0.1, solver="euler")
    backend.fastmath = True
    on = C.c_int(0)
    backend._lib.get_fastmath(C.byref(on))
    assert on.value == 1
    fast = _solve(backend, eq, state, t_range=20.0, dt=0.1, solver="euler")
    assert "fastmath" in backend._lib.last_kernel_name().decode()
    assert _rel(fast, exact) <= TOL
    assert not np.array_equal(fast, exact), "the contracted build produced the exact build's bits everywhere: is it really another build?"
    backend.fastmath = False
    again = _solve(backend, eq, state, t_range=20.0, dt=0.1, solver="euler")
    np.testing.assert_array_equal(again, exact)
    assert "fastmath" not in backend._lib.last_kernel_name().decode()


@pytest.mark.parametrize("case", ["cahn_hilliard_rk4", "expression_rkf45_f32", "cahn_hilliard_2d_euler", "operators"])
def test_other_paths_under_contraction(backend, case):
    rng = np.random.default_rng(4)
    if case == "operators":
        grid = pde_hip.CartesianGrid([[0, 3], [0, 2], [0, 4]], [48, 40, 128], periodic=[True, False, True])
        f = pde_hip.ScalarField(grid, rng.uniform(-1, 1, grid.shape))
        backend.fastmath = False
        ref = [f.laplace("auto_periodic_neumann", backend=backend).data, f.gradient("auto_periodic_neumann", backend=backend).data]
        backend.fastmath = True
        got = [f.laplace("auto_periodic_neumann", backend=backend).data, f.gradient("auto_periodic_neumann", backend=backend).data]
        for a, b in zip(got, ref):
            assert _rel(a, b) <= 1e-13
        return
    if case == "cahn_hilliard_rk4":
        grid, eq, kw, dtype, tol = pde_hip.UnitGrid((48, 48, 128), periodic=True), pde_hip.CahnHilliardPDE(1.0), dict(t_range=0.05, dt=1e-3, solver="runge-kutta"), np.float64, 1e-10
    elif case == "cahn_hilliard_2d_euler":
        grid, eq, kw, dtype, tol = pde_hip.UnitGrid((128, 128), periodic=True), pde_hip.CahnHilliardPDE(1.0), dict(t_range=0.2, dt=1e-3, solver="euler"), np.float64, 1e-10
    else:
        grid = pde_hip.UnitGrid((64, 64, 64), periodic=True)
        eq, kw, dtype, tol = pde_hip.PDE({"c": "laplace(c**3 - c - laplace(c)) - 0.01 * c"}), dict(t_range=0.05, solver="runge-kutta", adaptive=True), np.float32, 2e-6
    state = pde_hip.ScalarField(grid, rng.uniform(-0.1, 0.1, grid.shape), dtype=dtype)
    backend.fastmath = False
    exact = _solve(backend, eq, state, **kw)
    backend.fastmath = True
    fast = _solve(backend, eq, state, **kw)
    assert _rel(fast, exact) <= tol
