"""2-D nine-point Laplacian + corner ghost cells (pde/backends/numba/operators/cartesian.py:36-78, :153-190).

The stencil exists only in the reference's numba backend (not installable here), so the oracle is pinned on the
reference's KNOWN ANSWERS (tests/backends/numba_/operators/test_numba_cartesian_operators.py:203-242 and
tests/grids/test_cartesian_grids.py:311-321) and on an independent numpy restatement of the formulas; the HIP kernel
is then compared with the oracle bit for bit.
"""

from __future__ import annotations

import numpy as np
import pytest
from helpers import host_faces, oracle_grid, to_full

import pde_hip
from oracle import pde_oracle as O


def _numpy_laplace9(full, dx, w):
    """`value += arr[i+x-1, j+y-1] * stencil[x, y]` in the reference's loop order (cartesian.py:184-188)."""
    dxm2, dym2 = dx[0] ** -2.0, dx[1] ** -2.0
    dm2 = dxm2 + dym2
    st = np.array([[0.25 * dm2 * w, dxm2 * (1 - w), 0.25 * dm2 * w], [dym2 * (1 - w), (dxm2 + dym2) * (w - 2), dym2 * (1 - w)],
                   [0.25 * dm2 * w, dxm2 * (1 - w), 0.25 * dm2 * w]])
    nx, ny = full.shape[0] - 2, full.shape[1] - 2
    out = np.zeros((nx, ny))
    for x in range(3):
        for y in range(3):
            out = out + full[x : x + nx, y : y + ny] * st[x, y]
    return out


@pytest.mark.parametrize("periodic_x", [False, True])
@pytest.mark.parametrize("periodic_y", [False, True])
def test_corner_point_setter_known_answers(periodic_x, periodic_y):
    """The reference's own test of make_corner_point_setter_2d, on the oracle."""
    grid = pde_hip.UnitGrid([1, 1], periodic=[periodic_x, periodic_y])
    arr = np.array([[np.nan, 1, np.nan], [2, 3, 4], [np.nan, 5, np.nan]])
    if periodic_x:
        arr[0, :] = arr[2, :] = arr[1, :]
    if periodic_y:
        arr[:, 0] = arr[:, 2] = arr[:, 1]
    O.set_corner_points_2d(oracle_grid(grid), grid.periodic, arr)
    if periodic_x and periodic_y:
        np.testing.assert_allclose(arr, 3)
    elif periodic_x:
        np.testing.assert_allclose(arr, [[2, 3, 4], [2, 3, 4], [2, 3, 4]])
    elif periodic_y:
        np.testing.assert_allclose(arr, [[1, 1, 1], [3, 3, 3], [5, 5, 5]])
    else:
        np.testing.assert_allclose(2 * arr, [[3, 2, 5], [4, 6, 8], [7, 10, 9]])


def _field(grid, rng):
    x, y = grid.cell_coords[..., 0], grid.cell_coords[..., 1]
    return np.exp(-0.05 * ((x - 8) ** 2 + (y - 7) ** 2)) + 0.01 * rng.random(grid.shape)


@pytest.mark.parametrize("periodic", [[True, True], [True, False], [False, True], [False, False]])
def test_oracle_nine_point_stencil(periodic):
    """Known answers of the reference (5-point vs 9-point: equal for w -> 0, within 0.05 for w = 1/3 on a smooth field) and
    the independent numpy restatement (bit for bit: same products, same order)."""
    rng = np.random.default_rng(3)
    grid = pde_hip.CartesianGrid([[0, 16], [0, 12]], [16, 16], periodic=periodic)
    g = oracle_grid(grid)
    full = to_full(grid, _field(grid, rng))
    O.set_ghost_cells(g, 1, host_faces(grid.get_boundary_conditions("auto_periodic_neumann")).c, full)
    lap5 = O.laplace(g, full)
    lap_w0 = O.laplace9(g, grid.periodic, 1e-10, full.copy())
    np.testing.assert_allclose(lap5, lap_w0, rtol=1e-7, atol=1e-9)
    work = full.copy()
    lap_w3 = O.laplace9(g, grid.periodic, 1 / 3, work)
    np.testing.assert_allclose(lap5, lap_w3, atol=0.05)
    assert not np.array_equal(lap5, lap_w3)
    np.testing.assert_array_equal(lap_w3, _numpy_laplace9(work, grid.discretization, 1 / 3))   # corners were written into `work`
    # corners: periodic copies or the mean of the adjacent face ghosts
    if periodic[0]:
        assert work[0, 0] == work[-2, 0] and work[-1, -1] == work[1, -1]
    elif not periodic[1]:
        assert work[0, 0] == 0.5 * (work[0, 1] + work[1, 0]) and work[-1, -1] == 0.5 * (work[-1, -2] + work[-2, -1])


def test_reference_known_answer_gaussian():
    """tests/grids/test_cartesian_grids.py:311-321 on the oracle: exp(-x^2-y^2) on 17^2, |lap9 - lap5| <= w/3."""
    for periodic in (True, False):
        grid = pde_hip.CartesianGrid([[-1, 1], [-1, 1]], [17, 17], periodic=periodic)
        x, y = grid.cell_coords[..., 0], grid.cell_coords[..., 1]
        g = oracle_grid(grid)
        full = to_full(grid, np.exp(-x**2 - y**2))
        O.set_ghost_cells(g, 1, host_faces(grid.get_boundary_conditions("auto_periodic_neumann")).c, full)
        ref = O.laplace(g, full)
        for w in (1e-8, 1 / 3):
            np.testing.assert_allclose(ref, O.laplace9(g, grid.periodic, w, full.copy()), atol=w / 3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape,periodic", [((16, 16), [True, True]), ((33, 70), [False, True]), ((64, 257), [True, False]), ((5, 1024), [False, False])])
def test_hip_nine_point_matches_oracle(shape, periodic, dtype):
    """`field.laplace(bc, corner_weight=w, backend="hip")` == oracle, bit for bit (fp64 and fp32 storage)."""
    rng = np.random.default_rng(4)
    grid = pde_hip.CartesianGrid([[0, 3.0], [0, 5.0]], shape, periodic=periodic)
    data = rng.uniform(-1, 1, shape).astype(dtype)
    field = pde_hip.ScalarField(grid, data, dtype=dtype)
    bc = {"x": "periodic" if periodic[0] else {"value": 0.3}, "y": "periodic" if periodic[1] else {"derivative": -0.2}}
    g = oracle_grid(grid, dtype)
    for w in (0.5, 1 / 3):
        got = field.laplace(bc, corner_weight=w, backend="hip").data
        full = to_full(grid, data)
        O.set_ghost_cells(g, 1, host_faces(grid.get_boundary_conditions(bc)).c, full)
        np.testing.assert_array_equal(got, O.laplace9(g, grid.periodic, w, full))
    # w = 0 keeps the five-point kernel
    np.testing.assert_array_equal(field.laplace(bc, corner_weight=0, backend="hip").data, field.laplace(bc, backend="hip").data)
    # the operator with BCs (grid.make_operator) takes the keyword as well
    op = grid.make_operator("laplace", bc, backend="hip", corner_weight=0.5)
    full = to_full(grid, data)
    O.set_ghost_cells(g, 1, host_faces(grid.get_boundary_conditions(bc)).c, full)
    np.testing.assert_array_equal(op(data), O.laplace9(g, grid.periodic, 0.5, full))
