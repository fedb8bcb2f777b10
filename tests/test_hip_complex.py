"""GPU part of the complex-field work (CPU part: tests/test_complex.py): complex states as planar (re, im) pairs on
the MI355X through the mirror front end and the real ``libpdehip.so``.

* goldens recorded from the reference's solvers on complex arrays (tests/golden/make_golden_complex.py): Schroedinger with complex
  boundary values in 2-D and 3-D, a Gross-Pitaevskii-like equation with ``Abs`` / ``conjugate`` and a complex coefficient - Euler, RK4,
  adaptive RKF45, adaptive Euler: equal step counts, <= 1e-10;
* the same equation written as the real system of its parts (what the reference's own ``test_solvers_complex`` compares against):
  bit-identical on the device;
* ``pdehip_max_abs_pairs`` against the oracle; complex64 states; upload / download of complex host arrays.
"""

from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import pytest
from helpers import max_rel, oracle_grid, to_full

import pde_hip
from oracle import pde_oracle as O

pytestmark = pytest.mark.gpu

GOLD = np.load(Path(__file__).parent / "golden" / "complex.npz")
CASES = {c["id"]: c for c in json.loads(str(GOLD["cases"]))}
SOLVERS = [tuple(s) for s in json.loads(str(GOLD["solvers"]))]


def _c(v):
    return complex(v[0], v[1]) if isinstance(v, list) else v


def _equation(case):
    bc = {k: ({kk: _c(vv) for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in case["bc"].items()}
    consts = {k: _c(v) for k, v in case.get("consts", {}).items()}
    return pde_hip.PDE({case["var"]: case["rhs"]}, bc=bc, consts=consts)


@pytest.mark.parametrize("solver,adaptive", SOLVERS)
@pytest.mark.parametrize("cid", list(CASES))
def test_reference_goldens(cid, solver, adaptive):
    case = CASES[cid]
    grid = pde_hip.UnitGrid(case["shape"], periodic=case["periodic"])
    state = pde_hip.ScalarField(grid, GOLD[f"{cid}/input"])
    res, info = _equation(case).solve(state, t_range=case["t_range"], dt=case["dt"], solver=solver, adaptive=adaptive, ret_info=True, backend="hip")
    key = f"{cid}/{solver}{'_adaptive' if adaptive else ''}"
    got = np.array(res.data)
    assert got.dtype == np.complex128
    assert info["solver"]["steps"] == int(GOLD[f"{key}/steps"])
    assert max_rel(got, GOLD[f"{key}/final"]) < 1e-10


@pytest.mark.parametrize("solver", ["euler", "runge-kutta"])
def test_complex_equation_equals_the_real_system_of_its_parts(rng, solver):
    """`c = a + i b`, `dc/dt = -I laplace(c)`  <=>  `da/dt = laplace(b)`, `db/dt = -laplace(a)` (the yardstick of the reference's
    tests/solvers/test_generic_solvers.py:80-97): the same kernels on the same planar components - identical bits."""
    grid = pde_hip.UnitGrid([12, 10, 130], periodic=[True, False, True])
    a, b = rng.uniform(-1, 1, grid.shape), rng.uniform(-1, 1, grid.shape)
    def bc(value):
        return {"x": "periodic", "y": {"derivative": value}, "z": "periodic"}

    # the complex condition d_n c = 0.1 - 0.3 i: the real part belongs to the operand a, the imaginary part to the operand b - in the real
    # system the conditions are named after the EQUATION an operator stands in (pde/pdes/pde.py:232-264): laplace(b) is in a's equation
    res_c = pde_hip.PDE({"c": "-I * laplace(c)"}, bc=bc(0.1 - 0.3j)).solve(pde_hip.ScalarField(grid, a + 1j * b), t_range=0.02, dt=1e-3, solver=solver, backend="hip")
    res_r = pde_hip.PDE({"a": "laplace(b)", "b": "-laplace(a)"}, bc_ops={"a:laplace": bc(-0.3), "b:laplace": bc(0.1)}).solve(
        pde_hip.FieldCollection([pde_hip.ScalarField(grid, a), pde_hip.ScalarField(grid, b)]), t_range=0.02, dt=1e-3, solver=solver, backend="hip")
    got = np.array(res_c.data)
    np.testing.assert_array_equal(got.real, np.array(res_r.data)[0])
    np.testing.assert_array_equal(got.imag, np.array(res_r.data)[1])
    assert np.abs(got - (a + 1j * b)).max() > 1e-3


def test_axis_derivatives_and_gradient_squared_of_a_complex_field(rng):
    """`d_dx`, `d2_dy2`, `gradient_squared` of c = a + i b inside an expression (round 4): the split written out by hand as a real system
    of a and b - gs(c) = gs(a) - gs(b) + 2 i (d_x a d_x b + d_y a d_y b) - to rounding (another pass order), and against numpy central differences
    of the complex field."""
    grid = pde_hip.CartesianGrid([[0, 6], [0, 5]], [24, 40], periodic=True)
    a, b = rng.uniform(-1, 1, grid.shape), rng.uniform(-1, 1, grid.shape)
    eq_c = pde_hip.PDE({"c": "(0.5 + 0.2*I) * gradient_squared(c) - I * d_dx(c) + 0.3 * d2_dy2(c)"})
    eq_r = pde_hip.PDE({"a": "0.5 * (gradient_squared(a) - gradient_squared(b)) - 0.4 * (d_dx(a) * d_dx(b) + d_dy(a) * d_dy(b)) + d_dx(b) + 0.3 * d2_dy2(a)",
                        "b": "0.2 * (gradient_squared(a) - gradient_squared(b)) + 1.0 * (d_dx(a) * d_dx(b) + d_dy(a) * d_dy(b)) - d_dx(a) + 0.3 * d2_dy2(b)"})
    c0 = a + 1j * b
    res_c = eq_c.solve(pde_hip.ScalarField(grid, c0), t_range=0.01, dt=1e-3, solver="euler", backend="hip")
    res_r = eq_r.solve(pde_hip.FieldCollection([pde_hip.ScalarField(grid, a), pde_hip.ScalarField(grid, b)]), t_range=0.01, dt=1e-3, solver="euler", backend="hip")
    got = np.array(res_c.data)
    assert max_rel(got.real, np.array(res_r.data)[0]) < 1e-13 and max_rel(got.imag, np.array(res_r.data)[1]) < 1e-13
    dx, dy = (float(d) for d in grid.discretization)
    c = c0.copy()
    for _ in range(10):
        gx = (np.roll(c, -1, 0) - np.roll(c, 1, 0)) / (2 * dx)
        gy = (np.roll(c, -1, 1) - np.roll(c, 1, 1)) / (2 * dy)
        d2y = (np.roll(c, -1, 1) - 2 * c + np.roll(c, 1, 1)) / dy**2
        c = c + 1e-3 * ((0.5 + 0.2j) * (gx * gx + gy * gy) - 1j * gx + 0.3 * d2y)
    assert max_rel(got, c) < 1e-12


def test_vector_operators_of_a_complex_field(rng):
    """Round 5: `divergence` / `gradient` / `dot` of c = a + i b inside an expression against the real system written out by hand -
    -I div grad c = div grad b - I div grad a; dot(grad c, grad c) = sum |d_k c|^2 (the second operand is conjugated) - to rounding."""
    grid = pde_hip.CartesianGrid([[0, 6], [0, 5], [0, 4]], [12, 10, 136], periodic=True)
    a, b = rng.uniform(-1, 1, grid.shape), rng.uniform(-1, 1, grid.shape)
    eq_c = pde_hip.PDE({"c": "-I * divergence(gradient(c)) + 0.1 * dot(gradient(c), gradient(c))"})
    eq_r = pde_hip.PDE({"a": "divergence(gradient(b)) + 0.1 * (dot(gradient(a), gradient(a)) + dot(gradient(b), gradient(b)))", "b": "-divergence(gradient(a))"})
    res_c = eq_c.solve(pde_hip.ScalarField(grid, a + 1j * b), t_range=0.01, dt=1e-3, solver="runge-kutta", backend="hip")
    res_r = eq_r.solve(pde_hip.FieldCollection([pde_hip.ScalarField(grid, a), pde_hip.ScalarField(grid, b)]), t_range=0.01, dt=1e-3, solver="runge-kutta", backend="hip")
    got = np.array(res_c.data)
    assert max_rel(got.real, np.array(res_r.data)[0]) < 1e-13 and max_rel(got.imag, np.array(res_r.data)[1]) < 1e-13
    assert np.abs(got - (a + 1j * b)).max() > 1e-3


def test_a_real_state_turns_complex_and_complex64_stays_single(rng):
    grid = pde_hip.UnitGrid([16, 64], periodic=True)
    eq = pde_hip.PDE({"p": "I * laplace(p)"})
    assert eq.complex_valued
    y0 = rng.uniform(0, 1, grid.shape)
    res = eq.solve(pde_hip.ScalarField(grid, y0), t_range=0.01, dt=1e-3, solver="euler", backend="hip")        # pde/solvers/controller.py:430-432
    assert res.dtype == np.complex128 and np.abs(np.array(res.data).imag).max() > 1e-4
    res32 = eq.solve(pde_hip.ScalarField(grid, y0.astype(np.complex64)), t_range=0.01, dt=1e-3, solver="euler", backend="hip")
    assert res32.dtype == np.complex64
    assert max_rel(np.array(res32.data), np.array(res.data)) < 1e-5


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_modulus_norm_kernel_is_the_oracle(rng, dtype):
    from pde_hip.device import DeviceArray, DeviceScalar

    backend = pde_hip.get_backend("hip")
    grid = pde_hip.UnitGrid([6, 5, 131])
    info = backend.grid_info(grid, dtype)
    z = rng.normal(size=(3, *grid.shape)) + 1j * rng.normal(size=(3, *grid.shape))
    dev = DeviceArray(info, (3, 2), complex_pairs=True).set_valid(z, backend.stream)
    np.testing.assert_array_equal(dev.get_valid(stream=backend.stream), z.astype(dev.host_dtype))          # upload / download round trip
    err = DeviceScalar()
    backend._lib.max_abs_pairs(info.ref, 3, dev.ptr, err.ptr, backend.stream)
    planar = np.stack([z.real, z.imag], axis=1).reshape(6, *grid.shape).astype(dtype)
    assert err.value(backend.stream) == O.max_abs_pairs(oracle_grid(grid, dtype), 3, to_full(grid, planar))


def test_operators_on_complex_data(rng):
    grid = pde_hip.CartesianGrid([[0, 2], [0, 3], [0, 8]], [8, 9, 64], periodic=[False, True, True])
    z = rng.uniform(-1, 1, grid.shape) + 1j * rng.uniform(-1, 1, grid.shape)
    bc = {"x": {"value": 0.3 - 0.7j}, "y": "periodic", "z": "periodic"}
    backend = pde_hip.get_backend("hip")
    op = backend.make_operator(grid, "laplace", bcs=grid.get_boundary_conditions(bc, rank=0), dtype=complex)
    parts = []
    for value, take in ((0.3, np.real), (-0.7, np.imag)):
        bcp = {"x": {"value": value}, "y": "periodic", "z": "periodic"}
        parts.append(pde_hip.ScalarField(grid, take(z)).laplace(bcp, backend=backend).data)
    np.testing.assert_array_equal(op(z), parts[0] + 1j * parts[1])
