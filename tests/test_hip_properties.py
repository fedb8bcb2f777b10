"""Size-independent properties at the BASELINE sizes and at the sizes of the former tile cliff (GPU, whole field).

The oracle cannot run these grids in seconds; the stencil arithmetic offers properties that need no second implementation:

* **translation equivariance** on periodic grids, BIT FOR BIT: every cell sees the same operands in the same order wherever it
  sits, so stepping a rolled field equals rolling the stepped field.  The shifts move every cell across wave tiles, x-chunk seams,
  the moved last tiles of rows / columns that no tile divides (pdehip_march2.inc) and the periodic wrap;
* **exact scaling** by powers of two (the schemes are linear for diffusion; no rounding changes);
* **conservation** of the periodic sum (diffusion, Cahn-Hilliard) to rounding.

Reference semantics: pde/grids/operators/cartesian.py:220-227 (the stencil), pde/solvers/euler.py:172-175, pde/pdes/cahn_hilliard.py:115-122.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

import pde_hip
from pde_hip.device import DeviceArray, DeviceScalar, ptr_array

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def backend():
    return pde_hip.get_backend("hip")


def _euler(backend, eq, grid, data, dt, steps):
    spec = backend.make_rhs_spec(eq, pde_hip.ScalarField(grid, data, dtype=data.dtype))
    a, b = DeviceArray(spec.info).set_valid(data), DeviceArray(spec.info)
    res = C.c_void_p()
    backend._lib.euler_run(spec.info.ref, spec.ref, a.ptr, b.ptr, dt, steps, C.byref(res), None)
    return (b if res.value == b.ptr else a).get_valid()


@pytest.mark.parametrize("shape,dtype,shift", [
    ((512, 512, 512), np.float64, (131, 3, 77)),      # cfg4, the bench grid: across x-chunk seams (128 planes), row tiles, chunks
    ((513, 513, 513), np.float64, (1, 510, 385)),     # one cell in the last chunk, one row in the last tile
    ((511, 511, 511), np.float64, (300, 2, 129)),
    ((500, 500, 300), np.float64, (7, 499, 45)),      # the row ends inside the third chunk
    ((512, 512, 512), np.float32, (65, 1, 258)),
    ((257, 131, 259), np.float32, (5, 66, 3)),
    ((4095, 4097), np.float64, (2049, 130)),          # 2-D: the last chunk moved back by one cell
    ((1024, 1024), np.float64, (3, 515)),             # cfg2
])
def test_diffusion_steps_commute_with_translations(backend, shape, dtype, shift):
    """5 Euler steps (two two-step sweeps and a single step) of a rolled field == the rolled result, bit for bit; the field
    scaled by 2^-3 gives the result scaled by 2^-3, bit for bit; the sum is conserved."""
    grid = pde_hip.UnitGrid(shape, periodic=True)
    u = np.random.default_rng(7).random(shape).astype(dtype)
    eq = pde_hip.DiffusionPDE(1.0)
    dt = 0.1 if len(shape) == 3 else 0.2
    base = _euler(backend, eq, grid, u, dt, 5)
    moved = _euler(backend, eq, grid, np.roll(u, shift, axis=tuple(range(len(shape)))), dt, 5)
    np.testing.assert_array_equal(moved, np.roll(base, shift, axis=tuple(range(len(shape)))))
    del moved
    scaled = _euler(backend, eq, grid, (u * dtype(0.125)).astype(dtype), dt, 5)
    np.testing.assert_array_equal(scaled, base * dtype(0.125))
    tol = 1e-11 if dtype == np.float64 else 2e-5
    assert abs(base.sum(dtype=np.float64) - u.sum(dtype=np.float64)) < tol * u.sum(dtype=np.float64)
    assert base.min() > u.min() and base.max() < u.max()      # maximum principle


@pytest.mark.parametrize("shape,dtype,shift", [
    ((256, 256, 256), np.float32, (17, 130, 67)),     # cfg5
    ((512, 512), np.float64, (129, 258)),             # cfg3
    ((129, 67, 131), np.float64, (64, 1, 130)),       # moved last tiles along rows and row
])
def test_cahn_hilliard_runge_kutta_commutes_with_translations(backend, shape, dtype, shift):
    """One RKF45 attempt and three RK4 steps of Cahn-Hilliard (the fused two-level sweeps with their stage epilogues): rolled
    input == rolled output bit for bit, equal error estimates; the mass is conserved."""
    grid = pde_hip.UnitGrid(shape, periodic=True)
    u = np.random.default_rng(9).uniform(-0.5, 0.5, shape).astype(dtype)
    eq = pde_hip.CahnHilliardPDE(1.0)
    axes = tuple(range(len(shape)))
    lib = backend._lib
    results = []
    for data in (u, np.roll(u, shift, axis=axes)):
        spec = backend.make_rhs_spec(eq, pde_hip.ScalarField(grid, data, dtype=dtype))
        info = spec.info
        y, ynew, err = DeviceArray(info).set_valid(data), DeviceArray(info), DeviceScalar()
        work = [DeviceArray(info) for _ in range(7)]
        lib.rkf45_attempt(info.ref, spec.ref, y.ptr, ynew.ptr, ptr_array(work), 1e-3, err.ptr, None)
        attempt, e = ynew.get_valid(), err.value()
        for _ in range(3):
            lib.rk4_step(info.ref, spec.ref, y.ptr, ptr_array(work[:5]), 1e-3, None)
        results.append((attempt, e, y.get_valid()))
    (a0, e0, r0), (a1, e1, r1) = results
    assert e0 == e1 and e0 > 0
    np.testing.assert_array_equal(a1, np.roll(a0, shift, axis=axes))
    np.testing.assert_array_equal(r1, np.roll(r0, shift, axis=axes))
    tol = 1e-9 if dtype == np.float64 else 1e-3
    assert abs(r0.sum(dtype=np.float64) - u.sum(dtype=np.float64)) < tol * max(1.0, abs(u.sum(dtype=np.float64)))


def test_long_run_at_the_bench_size(backend):
    """1000 Euler steps of the bench workload (512^3 fp64, periodic, 500 two-step sweeps): the sum is conserved to rounding, the
    range shrinks monotonically towards the mean (maximum principle), and the result equals the same run done as 4 x 250
    steps - the loop has no state between calls - bit for bit."""
    n = 512
    grid = pde_hip.UnitGrid([n, n, n], periodic=True)
    u = np.random.default_rng(11).random((n, n, n))
    eq = pde_hip.DiffusionPDE(1.0)
    whole = _euler(backend, eq, grid, u, 0.1, 1000)
    total = u.sum(dtype=np.float64)
    assert abs(whole.sum(dtype=np.float64) - total) < 1e-10 * total
    assert u.min() < whole.min() and whole.max() < u.max()
    assert whole.max() - whole.min() < 0.05 * (u.max() - u.min())      # 1000 steps of dt = 0.1 smooth the white noise out
    part = u
    for _ in range(4):
        part = _euler(backend, eq, grid, part, 0.1, 250)
    np.testing.assert_array_equal(part, whole)
