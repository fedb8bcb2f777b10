"""Row lengths that are not a multiple of the 16-byte vector (the odd-size cliff).

The register-pipelined kernels now take ANY row length: the lane the row ends in computes its whole vector and stores
only the valid cells.  Everything here is compared with the oracle bit for bit — operators, on-the-fly BCs whose
virtual cell sits INSIDE a lane's vector, Euler loops, Runge-Kutta stage sweeps, expression kernels.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import pytest
from helpers import host_faces, interior, max_rel, oracle_grid, to_full
from test_oracle_golden import oracle_solve

import pde_hip
from oracle import pde_oracle as O
from pde_hip import _abi

pytestmark = pytest.mark.gpu

# (3-D rows 1..32 cells beyond whole 64-lane chunks are split into the aligned part and a strip: 129, 131, 145, 160, 257 / 130, 133, 259, 281, 288)
SHAPES64 = [(9, 7, 129), (5, 6, 131), (4, 9, 257), (11, 127), (8, 1025), (3, 5, 1), (6, 21, 145), (5, 40, 160)]
SHAPES32 = [(6, 5, 130), (3, 4, 133), (4, 6, 259), (7, 135), (5, 1027), (5, 19, 281), (4, 33, 288)]
BCS = {
    "periodic": lambda nd: "auto_periodic_neumann",
    "walls": lambda nd: {"x": {"value": 0.4}, "y": {"derivative": -0.3}, "z": {"type": "mixed", "value": 0.5, "const": 0.2}} if nd == 3
    else {"x": {"derivative": 0.2}, "y": {"value": -0.1}},
}


def _cases():
    for shape in SHAPES64:
        yield shape, np.float64
    for shape in SHAPES32:
        yield shape, np.float32


@pytest.mark.parametrize("bc_name", list(BCS))
@pytest.mark.parametrize("shape,dtype", list(_cases()))
def test_operators_any_row_length(shape, dtype, bc_name):
    nd = len(shape)
    periodic = bc_name == "periodic"
    grid = pde_hip.CartesianGrid([[0, 1.0 + 0.3 * a] for a in range(nd)], shape, periodic=periodic)
    bc = BCS[bc_name](nd)
    data = np.random.default_rng(1).uniform(-1, 1, shape).astype(dtype)
    field = pde_hip.ScalarField(grid, data, dtype=dtype)
    g = oracle_grid(grid, dtype)
    full = to_full(grid, data)
    O.set_ghost_cells(g, 1, host_faces(grid.get_boundary_conditions(bc)).c, full)
    np.testing.assert_array_equal(field.laplace(bc, backend="hip").data, O.laplace(g, full))
    for method in ("central", "forward", "backward"):
        np.testing.assert_array_equal(field.gradient(bc, backend="hip", method=method).data, O.gradient(g, full, method))
    for central in (True, False):
        np.testing.assert_array_equal(field.gradient_squared(bc, backend="hip", central=central).data, O.gradient_squared(g, full, central))


@pytest.mark.parametrize("kind", ["diffusion", "cahn_hilliard"])
@pytest.mark.parametrize("solver,dt", [("euler", 1e-3), ("runge-kutta", 1e-3), ("runge-kutta", None)])
@pytest.mark.parametrize("shape,dtype", [((9, 7, 129), np.float64), ((6, 5, 130), np.float32), ((11, 127), np.float64),
                                         ((6, 9, 257), np.float64), ((5, 7, 261), np.float32), ((9, 131), np.float64)])
def test_steppers_any_row_length(shape, dtype, kind, solver, dt):
    """Euler loop (BCs on the fly incl. the upper face of the fastest axis inside a lane's vector; rows longer than a chunk: the
    two-level kernel with its last tiles moved back over their neighbours), RK4 and RKF45 stage sweeps with their pointwise
    streams: equal step counts, bit-identical to the oracle (fp32: 1e-6)."""
    nd = len(shape)
    grid = pde_hip.UnitGrid(shape, periodic=[True] + [False] * (nd - 1))
    bc = "auto_periodic_neumann" if kind == "cahn_hilliard" else {"x": "periodic", "y": {"value": 0.3}, **({"z": {"derivative": 0.1}} if nd == 3 else {})}
    data = np.random.default_rng(2).uniform(-0.3, 0.3, shape).astype(dtype)
    eq = pde_hip.DiffusionPDE(0.8, bc=bc) if kind == "diffusion" else pde_hip.CahnHilliardPDE(0.9, bc_c=bc, bc_mu=bc)
    t_range = 7e-3 if dt else 0.02
    res, info = eq.solve(pde_hip.ScalarField(grid, data, dtype=dtype), t_range=t_range, dt=dt, solver=solver, backend="hip", ret_info=True)
    case = {"pde": kind, "D": 0.8, "gamma": 0.9, "bc": bc, "t_range": t_range, "dt": dt, "solver": solver}
    expect, steps, _ = oracle_solve(case, grid, dtype, data)
    assert info["solver"]["steps"] == steps
    if dtype == np.float32:
        assert max_rel(res.data.astype(np.float64), expect.astype(np.float64)) < 1e-6
    else:
        np.testing.assert_array_equal(res.data, expect)


def test_expression_kernels_any_row_length():
    """Run-time compiled epilogues ride on the same kernel: an Allen-Cahn rate on an odd row length."""
    grid = pde_hip.UnitGrid((7, 6, 129), periodic=[True, False, True])
    data = np.random.default_rng(3).uniform(-1, 1, grid.shape)
    state = pde_hip.ScalarField(grid, data)
    eq = pde_hip.PDE({"c": "c - c**3 + laplace(c)"})
    rate = eq.evolution_rate(state).data
    lap = state.laplace("auto_periodic_neumann", backend="hip").data
    assert max_rel(rate, data - data**3 + lap) < 1e-14


@pytest.mark.parametrize("n", [511, 513])
def test_odd_cubes_full_field(n):
    """511^3 / 513^3 (the sizes the cliff was measured at): 3 Euler steps, whole field, bit-identical to the oracle."""
    grid = pde_hip.UnitGrid([n, n, n], periodic=True)
    u = np.random.default_rng(0).random((n, n, n))
    res = pde_hip.DiffusionPDE().solve(pde_hip.ScalarField(grid, u), t_range=0.3, dt=0.1, solver="euler", backend="hip")
    g = oracle_grid(grid)
    rhs = O.make_rhs(_abi.RHS_DIFFUSION, 1.0, host_faces(grid.get_boundary_conditions("auto_periodic_neumann")).c)
    np.testing.assert_array_equal(res.data, interior(grid, O.euler_run(g, rhs, to_full(grid, u), 0.1, 3)))


OPEN_CASES = {
    # shape, periodic, bc: fields of 8 M cells and more whose row COUNT is one to four beyond a whole number of tiles ("open" columns of tiles,
    # round 6: launch_euler2_tv) - alone, together with open rows (columns beyond whole chunks), with local faces on either axis
    "walls-1-row-8-columns": ((40, 513, 520), False, {"x": {"value": 0.2}, "y-": {"derivative": 0.1}, "y+": {"value": -0.3}, "z-": {"type": "mixed", "value": 0.5, "const": 0.1}, "z+": {"derivative": -0.2}}),
    "periodic-rows-3": ((72, 323, 384), [False, True, False], {"x": {"derivative": 0.1}, "y": "periodic", "z": {"value": 0.4}}),
    "walls-2-rows-1-column": ((64, 518, 257), False, {"x": {"value": 0.0}, "y": {"value": 0.5}, "z-": {"derivative": 0.2}, "z+": {"value": -0.1}}),
    "all-periodic-tall-4-rows": ((224, 516, 512), True, {}),
    "all-periodic-tall-7-rows-2-columns": ((200, 519, 514), True, {}),
    "all-periodic-tall-5-rows": ((224, 517, 512), True, {}),
    "walls-tall-forced-sizes": ((224, 515, 512), False, {"x": {"value": 0.1}, "y-": {"value": 0.3}, "y+": {"derivative": -0.4}, "z": {"derivative": 0.0}}),
}


@pytest.mark.parametrize("case", sorted(OPEN_CASES))
def test_open_rows_and_open_tile_columns_full_field(case):
    """4 Euler steps (two sweeps + the recomputed rows / columns behind them), every cell against the oracle."""
    shape, periodic, bc = OPEN_CASES[case]
    grid = pde_hip.UnitGrid(shape, periodic=periodic)
    bcs = bc if bc else "auto_periodic_neumann"
    u = np.random.default_rng(3).uniform(-1, 1, shape)
    res = pde_hip.DiffusionPDE(0.7, bc=bcs).solve(pde_hip.ScalarField(grid, u), t_range=0.4, dt=0.1, solver="euler", backend="hip", tracker=None)
    g = oracle_grid(grid)
    rhs = O.make_rhs(_abi.RHS_DIFFUSION, 0.7, host_faces(grid.get_boundary_conditions(bcs)).c)
    np.testing.assert_array_equal(res.data, interior(grid, O.euler_run(g, rhs, to_full(grid, u), 0.1, 4)))


@pytest.mark.parametrize("shape", [(96, 513, 513), (80, 515, 520)])
def test_open_rows_and_open_tile_columns_fp32(shape):
    """fp32 all-periodic: the wide 4-row tile over a whole number of tiles and chunks, the rows / columns behind them recomputed (round 6) - every
    cell bit-identical to the oracle's fp32 loop (fp32 storage, fp64 registers)."""
    grid = pde_hip.UnitGrid(shape, periodic=True)
    data = np.random.default_rng(4).uniform(-1, 1, shape).astype(np.float32)
    res, info = pde_hip.DiffusionPDE(0.8).solve(pde_hip.ScalarField(grid, data, dtype=np.float32), t_range=0.4, dt=0.1, solver="euler", backend="hip", ret_info=True, tracker=None)
    case = {"pde": "diffusion", "D": 0.8, "gamma": 0.0, "bc": "auto_periodic_neumann", "t_range": 0.4, "dt": 0.1, "solver": "euler"}
    expect, steps, _ = oracle_solve(case, grid, np.float32, data)
    assert info["solver"]["steps"] == steps
    np.testing.assert_array_equal(res.data, expect)
