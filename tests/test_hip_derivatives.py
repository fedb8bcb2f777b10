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
