"""GPU parity of the register-pipelined gradient / gradient_squared / divergence kernels (and their
generic fallbacks) against the CPU oracle on seeded inputs, bit-exact, for tile geometries that
exercise chunk boundaries, partially filled tiles, single-cell axes and both dtypes."""

from __future__ import annotations

import numpy as np
import pytest
from helpers import host_faces, oracle_grid, to_full

import pde_hip
from oracle import pde_oracle as O
from pde_hip import _abi
from pde_hip.device import DeviceArray, DeviceBuffer

pytestmark = pytest.mark.gpu

SHAPES = [(6, 10, 128), (5, 7, 200), (4, 6, 520), (3, 3, 1032), (2, 5, 256), (9, 256), (5, 600), (33, 130), (2, 2, 4), (1, 1, 8),
          (12, 16, 64), (7, 9, 11), (40,), (64, 64, 64)]


@pytest.fixture(scope="module")
def backend():
    return pde_hip.get_backend("hip")


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("layout", [_abi.OUT_FULL, _abi.OUT_VALID])
def test_derivative_family_vs_oracle(backend, shape, dtype, layout):
    nd = len(shape)
    grid = pde_hip.CartesianGrid([[0, n * (0.7 + 0.2 * a)] for a, n in enumerate(shape)], shape, periodic=[a % 2 == 0 for a in range(nd)])
    rng = np.random.default_rng(17)
    data = rng.uniform(-1, 1, shape).astype(dtype)
    vdata = rng.uniform(-1, 1, (nd, *shape)).astype(dtype)
    g = oracle_grid(grid, dtype)
    bcs = grid.get_boundary_conditions("auto_periodic_neumann", rank=0)
    vbcs = grid.get_boundary_conditions("auto_periodic_neumann", rank=1)
    full = to_full(grid, data)
    O.set_ghost_cells(g, 1, host_faces(bcs).c, full)
    vfull = to_full(grid, vdata)
    O.set_ghost_cells(g, nd, host_faces(vbcs, (nd,)).c, vfull)

    info = backend.grid_info(grid, dtype)
    lib = backend._lib
    dev = DeviceArray(info).set_valid(data)
    vdev = DeviceArray(info, (nd,)).set_valid(vdata)
    backend.make_ghost_cell_setter(bcs)(dev)
    backend.make_ghost_cell_setter(vbcs)(vdev)
    cells = int(np.prod(shape))

    def run(call, ncomp_out):
        """Run an operator into a full or a valid device array and return host valid data."""
        if layout == _abi.OUT_FULL:
            out = DeviceArray(info, (ncomp_out,) if ncomp_out > 1 else ())
            call(out.ptr)
            return out.get_valid()
        buf = DeviceBuffer(cells * ncomp_out * np.dtype(dtype).itemsize)
        call(buf.ptr)
        host = np.empty(((ncomp_out,) if ncomp_out > 1 else ()) + tuple(shape), dtype)
        lib.memcpy_d2h(host.ctypes.data, buf.ptr, host.nbytes, None)
        return host

    for method, code in _abi.METHODS.items():
        got = run(lambda p: lib.gradient(info.ref, code, dev.ptr, p, layout, None), nd)
        np.testing.assert_array_equal(got.reshape((nd, *shape)), O.gradient(g, full, method), err_msg=f"gradient {method}")
        got = run(lambda p: lib.divergence(info.ref, code, vdev.ptr, p, layout, None), 1)
        np.testing.assert_array_equal(got, O.divergence(g, vfull, method), err_msg=f"divergence {method}")
    for central in (True, False):
        got = run(lambda p: lib.gradient_squared(info.ref, int(central), dev.ptr, p, layout, None), 1)
        np.testing.assert_array_equal(got, O.gradient_squared(g, full, central), err_msg=f"gradient_squared {central}")


def test_div_grad_equals_wide_laplacian(backend):
    """div(grad u) with central differences is the 2h-Laplacian: constant -> 0, x^2 -> 2 away from the walls
    (tests/backends/numba_/operators/test_numba_cartesian_operators.py:118-170)."""
    grid = pde_hip.CartesianGrid([[0, 1], [0, 1]], [64, 64], periodic=True)
    u = pde_hip.ScalarField.from_expression(grid, "sin(2*pi*x) * cos(2*pi*y)")
    grad = u.gradient("periodic")
    div = grad.divergence("periodic")
    lap = u.laplace("periodic")
    np.testing.assert_allclose(div.data, lap.data, atol=0.05 * np.abs(lap.data).max())
    const = pde_hip.ScalarField(grid, 2.5)
    np.testing.assert_array_equal(const.gradient("periodic").data, 0)
    np.testing.assert_array_equal(const.gradient_squared("periodic").data, 0)


@pytest.mark.parametrize("shape", [(7, 9, 12), (33, 130), (40,)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_axis_derivative_pattern_operators(backend, shape, dtype):
    """`d_dx`, `d_dy_forward`, `d2_dz2`, ... (numba/backend.py:143-173) vs the oracle (bit-exact) and, in 1-D,
    vs the reference's own gradient (whose 1-D formula is the same `/ (2 dx)` expression)."""
    nd = len(shape)
    grid = pde_hip.CartesianGrid([[0, n * 0.6] for n in shape], shape)
    data = np.random.default_rng(4).uniform(-1, 1, shape).astype(dtype)
    field = pde_hip.ScalarField(grid, data, dtype=dtype)
    bc = {"value": 0.2}
    g = oracle_grid(grid, dtype)
    full = to_full(grid, data)
    O.set_ghost_cells(g, 1, host_faces(grid.get_boundary_conditions(bc)).c, full)
    for ax, name in enumerate(grid.axes):
        np.testing.assert_array_equal(field.apply_operator(f"d_d{name}", bc).data, O.axis_derivative(g, full, ax, 1, "central"))
        np.testing.assert_array_equal(field.apply_operator(f"d_d{name}_forward", bc).data, O.axis_derivative(g, full, ax, 1, "forward"))
        np.testing.assert_array_equal(field.apply_operator(f"d_d{name}_backward", bc).data, O.axis_derivative(g, full, ax, 1, "backward"))
        np.testing.assert_array_equal(field.apply_operator(f"d2_d{name}2", bc).data, O.axis_derivative(g, full, ax, 2))
    if nd == 1:
        np.testing.assert_array_equal(field.apply_operator("d_dx", bc).data, field.gradient(bc).data[0])
    # the sum of the second derivatives is the Laplacian up to rounding (tests/grids/test_cartesian_grids.py:276-294)
    total = sum(field.apply_operator(f"d2_d{name}2", bc).data.astype(np.float64) for name in grid.axes)
    np.testing.assert_allclose(total, field.laplace(bc).data, rtol=1e-5 if dtype == np.float32 else 1e-12, atol=1e-4 if dtype == np.float32 else 1e-10)
    with pytest.raises(NotImplementedError, match="does not define operator"):
        field.apply_operator("d_dq", bc)
