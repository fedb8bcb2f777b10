"""Pin the CPU oracle (oracle/pde_oracle.c) against the reference's own outputs (CPU only).

Golden vectors come from tests/golden/make_golden.py (reference numpy ghost cells, eager
torch-CPU operators + Euler stepper, scipy operators, numpy+scipy RK solvers).  Tolerances:
  * ghost cells vs reference numpy                  : bit-exact
  * fp64 operators / Euler steppers vs torch-CPU    : bit-exact (same expression order)
  * fp64 operators vs scipy (ndimage)               : rtol 1e-12 (scipy sums in another order)
  * RK4 / RKF45 / adaptive Euler vs numpy+scipy     : rel 1e-10 on the field, same step count
  * fp32 vs torch (pure fp32 arithmetic)            : rel 2e-6 (oracle computes in fp64 registers)
"""

from __future__ import annotations

import json

import numpy as np
import pytest
from helpers import case_ids, face_mask, get_case, host_faces, interior, make_grid, max_rel, oracle_grid, to_full

import pde_hip
from oracle import pde_oracle as O
from pde_hip import _abi
from pde_hip.solvers import OnlineStatistics, make_dt_adjuster

OPS = case_ids("ops.npz")
STEPS = case_ids("steppers.npz")


def _setup(golden_ops, cid):
    case = get_case(golden_ops, cid)
    grid = make_grid(case)
    dtype = np.dtype(case.get("dtype", "float64"))
    g = oracle_grid(grid, dtype)
    return case, grid, dtype, g


@pytest.mark.parametrize("cid", OPS)
def test_ghost_cells_bit_exact(golden_ops, cid):
    case, grid, dtype, g = _setup(golden_ops, cid)
    bcs = grid.get_boundary_conditions(case["bc"], rank=0)
    full = to_full(grid, golden_ops[f"{cid}/input"])
    O.set_ghost_cells(g, 1, host_faces(bcs).c, full)
    ref = golden_ops[f"{cid}/full"]
    mask = face_mask(grid)
    np.testing.assert_array_equal(full[mask], ref[mask])
    # vector field, default BCs
    vbcs = grid.get_boundary_conditions("auto_periodic_neumann", rank=1)
    vfull = to_full(grid, golden_ops[f"{cid}/vector_input"])
    O.set_ghost_cells(g, grid.dim, host_faces(vbcs, (grid.dim,)).c, vfull)
    vmask = face_mask(grid, (grid.dim,))
    np.testing.assert_array_equal(vfull[vmask], golden_ops[f"{cid}/vector_full"][vmask])


def test_normal_bc(golden_ops):
    grid = pde_hip.UnitGrid([4, 5], periodic=[False, True])
    bc = json.loads(str(golden_ops["normal_bc/bc"]))
    bcs = grid.get_boundary_conditions(bc, rank=1)
    full = to_full(grid, golden_ops["normal_bc/input"])
    O.set_ghost_cells(oracle_grid(grid), 2, host_faces(bcs, (2,)).c, full)
    ref = golden_ops["normal_bc/full"]
    # the tangential component of a normal BC is left untouched (zero here, garbage in the reference)
    mask = face_mask(grid, (2,)).copy()
    mask[1, 0, :] = mask[1, -1, :] = False
    np.testing.assert_array_equal(full[mask], ref[mask])


def _check(out, ref, dtype, exact=True, rtol=1e-12):
    if dtype == np.float32:
        assert max_rel(out.astype(np.float64), ref.astype(np.float64)) < 2e-6
    elif exact:
        np.testing.assert_array_equal(out, ref)
    else:
        np.testing.assert_allclose(out, ref, rtol=rtol, atol=rtol * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("cid", OPS)
def test_scalar_operators(golden_ops, cid):
    case, grid, dtype, g = _setup(golden_ops, cid)
    full = golden_ops[f"{cid}/full"].copy()
    _check(O.laplace(g, full), golden_ops[f"{cid}/laplace_torch"], dtype)
    if f"{cid}/laplace_scipy" in golden_ops:
        _check(O.laplace(g, full), golden_ops[f"{cid}/laplace_scipy"], dtype, exact=False)
    _check(O.gradient(g, full, "central"), golden_ops[f"{cid}/gradient_central_torch"], dtype, exact=grid.dim > 1, rtol=1e-14)
    for method in ["central", "forward", "backward"]:
        key = f"{cid}/gradient_{method}_scipy"
        if key in golden_ops:
            _check(O.gradient(g, full, method), golden_ops[key], dtype, exact=False)
    _check(O.gradient_squared(g, full, True), golden_ops[f"{cid}/gradient_squared_central_torch"], dtype, exact=False, rtol=1e-14)
    key = f"{cid}/gradient_squared_noncentral_torch"
    if key in golden_ops:
        _check(O.gradient_squared(g, full, False), golden_ops[key], dtype, exact=False, rtol=1e-14)
    # laplace into a full-layout output: interior identical, ghosts untouched (zero)
    out_full = O.laplace(g, full, _abi.OUT_FULL)
    np.testing.assert_array_equal(interior(grid, out_full), O.laplace(g, full))


@pytest.mark.parametrize("cid", OPS)
def test_vector_operators(golden_ops, cid):
    case, grid, dtype, g = _setup(golden_ops, cid)
    vfull = golden_ops[f"{cid}/vector_full"].copy()
    _check(O.divergence(g, vfull, "central"), golden_ops[f"{cid}/divergence_central_torch"], dtype, exact=False, rtol=1e-14)
    for method in ["central", "forward", "backward"]:
        key = f"{cid}/divergence_{method}_scipy"
        if key in golden_ops:
            _check(O.divergence(g, vfull, method), golden_ops[key], dtype, exact=False)
    vlap = np.stack([O.laplace(g, np.ascontiguousarray(vfull[i])) for i in range(grid.dim)])
    _check(vlap, golden_ops[f"{cid}/vector_laplace_torch"], dtype)
    vgrad = np.stack([O.gradient(g, np.ascontiguousarray(vfull[i])) for i in range(grid.dim)])
    _check(vgrad, golden_ops[f"{cid}/vector_gradient_torch"], dtype, exact=grid.dim > 1, rtol=1e-14)


def test_known_answers():
    """Reference known-answer tests (tests/backends/numba_/operators/test_numba_cartesian_operators.py:118-170)."""
    grid = pde_hip.CartesianGrid([[0, 2 * np.pi]] * 2, 16, periodic=True)
    g = oracle_grid(grid)
    x, y = grid.cell_coords[..., 0], grid.cell_coords[..., 1]
    bcs = grid.get_boundary_conditions("auto_periodic_neumann")
    # harmonic field: laplace(s) ~ -s  (tests/fields/test_scalar_fields.py:108-120)
    s = np.sin(x) + np.cos(y)
    full = to_full(grid, s)
    O.set_ghost_cells(g, 1, host_faces(bcs).c, full)
    np.testing.assert_allclose(O.laplace(g, full), -s, rtol=0.1, atol=0.1)
    # constant -> 0, gradient of constant -> 0
    full = to_full(grid, np.full(grid.shape, 3.0))
    O.set_ghost_cells(g, 1, host_faces(bcs).c, full)
    np.testing.assert_allclose(O.laplace(g, full), 0, atol=1e-10)
    np.testing.assert_allclose(O.gradient(g, full), 0, atol=1e-10)
    # x**2 -> 2 in the interior of a non-periodic grid with extrapolating BCs
    grid = pde_hip.CartesianGrid([[0, 1]], 8)
    g = oracle_grid(grid)
    xs = grid.axes_coords[0]
    full = to_full(grid, xs**2)
    O.set_ghost_cells(g, 1, host_faces(grid.get_boundary_conditions({"curvature": 2.0})).c, full)
    np.testing.assert_allclose(O.laplace(g, full), 2.0, rtol=1e-10)


def _rhs_from_case(case, grid, dtype, scratch):
    bcs = grid.get_boundary_conditions(case["bc"], rank=0)
    faces = host_faces(bcs)
    if case["pde"] == "diffusion":
        return O.make_rhs(_abi.RHS_DIFFUSION, case["D"], faces.c), faces
    return O.make_rhs(_abi.RHS_CAHN_HILLIARD, case["gamma"], faces.c, faces.c, scratch), faces


def oracle_solve(case, grid, dtype, state_valid):
    """Drive the oracle kernels with the same controller logic the product uses."""
    g = oracle_grid(grid, dtype)
    y = to_full(grid, state_valid.astype(dtype))
    scratch = np.zeros_like(y)
    rhs, _keep = _rhs_from_case(case, grid, dtype, scratch)
    t_end = case["t_range"]
    if case["dt"] is not None:
        dt = case["dt"]
        steps = max(1, round(t_end / dt))
        if case["solver"] == "euler":
            y = O.euler_run(g, rhs, y, dt, steps)
        elif case["solver"] == "adams-bashforth":
            y = O.adams_bashforth_run(g, rhs, y, dt, steps)
        else:
            for _ in range(steps):
                O.rk4_step(g, rhs, y, dt)
        return interior(grid, y), steps, dt
    # adaptive loop: pde/backends/numba/_solvers.py:249-281
    adjust = make_dt_adjuster(1e-10, 1e10)
    stats = OnlineStatistics()
    dt_opt, t, steps, tol = 1e-3, 0.0, 0, 1e-4
    while True:
        dt_step = max(min(dt_opt, t_end - t), 1e-10)
        if case["solver"] == "runge-kutta":
            ynew, err = O.rkf45_attempt(g, rhs, y, dt_step)
        else:
            k1 = O.euler_run(g, rhs, y, dt_step, 1)
            k2a = O.euler_run(g, rhs, y, 0.5 * dt_step, 1)
            ynew = O.euler_run(g, rhs, k2a, 0.5 * dt_step, 1)
            err = O.max_abs_diff(g, 1, k1, ynew)
        error_rel = err / tol
        if error_rel <= 1:
            steps += 1
            t += dt_step
            y = ynew
            stats.add(dt_step)
        if t < t_end:
            dt_opt = adjust(dt_step, error_rel)
        else:
            break
    return interior(grid, y), steps, dt_opt


@pytest.mark.parametrize("cid", STEPS)
def test_steppers(golden_steppers, cid):
    case = get_case(golden_steppers, cid)
    grid = make_grid(case)
    dtype = np.dtype(case.get("dtype", "float64"))
    final, steps, dt_last = oracle_solve(case, grid, dtype, golden_steppers[f"{cid}/input"])
    ref = golden_steppers[f"{cid}/final"]
    assert steps == int(golden_steppers[f"{cid}/steps"])
    if dtype == np.float32:
        assert max_rel(final.astype(np.float64), ref.astype(np.float64)) < 1e-5
    elif case["backend"] == "torch":
        np.testing.assert_array_equal(final, ref)  # same expression order as torch-CPU: bit-exact
    else:
        assert max_rel(final, ref) < 1e-10
        np.testing.assert_allclose(dt_last, float(golden_steppers[f"{cid}/dt_last"]), rtol=1e-6)
