"""Euler-Maruyama noise increments (pde/solvers/euler.py:66-147).

The device generator (Philox4x32-10 counter-based + Box-Muller, ``pdehip_add_gaussian_noise``) has a CPU twin in the
oracle; the integer part is identical, the transcendental part (log, cos, sqrt) differs by the usual last-bit
differences between device and host libm, so HIP is compared with the oracle at 1e-12 (absolute, unit variance).
Statistics (mean, variance, Kolmogorov-Smirnov against N(0, 1), independence of successive calls) pin the generator
itself — the reference's backends draw from numba's / torch's generators, so realisations are never comparable, only
distributions are (tests/pdes/test_diffusion_pdes.py:95-109, run with the hip backend in tests/test_reference_suite.py).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import pytest
from scipy import stats

import pde_hip
from oracle import pde_oracle as O
from pde_hip import _abi


def _oracle_noise(shape, scale, seed, counter, offset=0, dtype=np.float64, base=0.0):
    lib = O.lib()
    lib.oracle_add_gaussian_noise.argtypes = [C.POINTER(_abi.Grid), C.c_int, C.c_void_p, C.c_double, C.c_uint64, C.c_uint64, C.c_uint64]
    g = _abi.make_grid(shape, (1.0,) * len(shape), dtype)
    full = np.full(tuple(s + 2 for s in shape), base, dtype=dtype)
    assert lib.oracle_add_gaussian_noise(C.byref(g), 1, full.ctypes.data, scale, seed, counter, offset) == 0
    return full[(slice(1, -1),) * len(shape)].copy()


def test_generator_statistics_and_reproducibility():
    x = _oracle_noise((64, 64, 32), 1.0, seed=1234, counter=0)
    assert abs(x.mean()) < 4 / np.sqrt(x.size) and abs(x.var() - 1) < 0.02
    assert stats.kstest(x.ravel(), stats.norm().cdf).pvalue > 1e-3
    np.testing.assert_array_equal(x, _oracle_noise((64, 64, 32), 1.0, seed=1234, counter=0))      # same seed, same call: same field
    y = _oracle_noise((64, 64, 32), 1.0, seed=1234, counter=1)                                     # next call: independent field
    assert abs(np.corrcoef(x.ravel(), y.ravel())[0, 1]) < 0.02
    z = _oracle_noise((64, 64, 32), 1.0, seed=1235, counter=0)
    assert abs(np.corrcoef(x.ravel(), z.ravel())[0, 1]) < 0.02
    # neighbouring cells are uncorrelated
    assert abs(np.corrcoef(x[:, :, :-1].ravel(), x[:, :, 1:].ravel())[0, 1]) < 0.02
    # a slab of the grid draws the same numbers as the same cells of the whole grid (cell_offset = first global cell)
    lower = _oracle_noise((32, 64, 32), 1.0, seed=1234, counter=0, offset=0)
    upper = _oracle_noise((32, 64, 32), 1.0, seed=1234, counter=0, offset=32 * 64 * 32)
    np.testing.assert_array_equal(np.concatenate([lower, upper]), x)
    # scale and base value: y += scale * xi
    np.testing.assert_allclose(_oracle_noise((64, 64, 32), 0.5, 1234, 0, base=2.0), 2.0 + 0.5 * x, rtol=0, atol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dtype", [((16, 12, 128), np.float64), ((37, 129), np.float64), ((100,), np.float64), ((8, 8, 64), np.float32)])
def test_hip_noise_matches_the_cpu_twin(shape, dtype):
    from pde_hip.device import DeviceArray

    backend = pde_hip.get_backend("hip")
    grid = pde_hip.UnitGrid(shape)
    info = backend.grid_info(grid, dtype)
    base = np.full(shape, 0.25, dtype)
    arr = DeviceArray(info).set_valid(base)
    backend._lib.add_gaussian_noise(info.ref, 1, arr.ptr, 0.7, 99, 5, 11, None)
    expect = _oracle_noise(shape, 0.7, 99, 5, offset=11, dtype=dtype, base=0.25)
    got = arr.get_valid()
    if dtype == np.float32:
        np.testing.assert_allclose(got, expect, rtol=0, atol=1e-6)
    else:
        np.testing.assert_allclose(got, expect, rtol=0, atol=1e-12)
    assert abs((got - 0.25).std() / 0.7 - 1) < 0.05


@pytest.mark.gpu
def test_euler_maruyama_noise_scaling():
    """tests/pdes/test_diffusion_pdes.py:95-109 through the mirror API: D = 0, so c(t) ~ N(0, noise * t / dx) per cell."""
    var_local, t_range = 0.35, 0.1
    grid = pde_hip.CartesianGrid([[0, 1000]], 3700)
    eq = pde_hip.DiffusionPDE(0, noise=var_local, rng=np.random.default_rng(0))
    sol, info = eq.solve(pde_hip.ScalarField(grid), t_range=t_range, dt=1e-4, solver="euler", backend="hip", ret_info=True)
    assert info["solver"]["stochastic"] and info["solver"]["steps"] == 1000
    var_expected = var_local * t_range / grid.discretization[0]
    assert stats.kstest(np.ravel(sol.data), stats.norm(scale=np.sqrt(var_expected)).cdf).pvalue > 0.01
    # diffusion + noise in 3-D stays finite and the mean follows the deterministic equation (zero-mean increments)
    grid3 = pde_hip.UnitGrid([16, 16, 64], periodic=True)
    state = pde_hip.ScalarField(grid3, 1.0)
    sol3 = pde_hip.DiffusionPDE(1.0, noise=1e-2, rng=np.random.default_rng(1)).solve(state, t_range=1.0, dt=0.05, solver="euler", backend="hip")
    assert np.isfinite(sol3.data).all() and abs(sol3.data.mean() - 1) < 5e-3 and sol3.data.std() > 1e-3
    with pytest.raises(RuntimeError, match="adaptive stepping with stochastic"):
        pde_hip.DiffusionPDE(1.0, noise=1e-2).solve(state, t_range=0.1, dt=None, solver="euler", backend="hip")


@pytest.mark.gpu
def test_per_field_noise_of_a_collection():
    """`PDE({"a": 0, "b": 0}, noise=[va, vb])`: every field of the collection gets its own variance and its own random stream
    (pde/pdes/pde.py:266-281; the reference's KS test of the same set-up runs in tests/test_reference_suite.py: test_pde_noise)."""
    from scipy import stats

    import pde_hip

    grid = pde_hip.UnitGrid([128, 128])
    zero = np.zeros(grid.shape)
    state = pde_hip.FieldCollection([pde_hip.ScalarField(grid, zero), pde_hip.ScalarField(grid, zero)])
    eq = pde_hip.PDE({"a": "0*a", "b": "0*b"}, noise=[0.01, 2.0], rng=np.random.default_rng(3))
    res = eq.solve(state, t_range=1, dt=1, solver="euler", backend="hip")
    a, b = res.data[0].ravel(), res.data[1].ravel()
    assert stats.kstest(a, stats.norm(scale=np.sqrt(0.01)).cdf).pvalue > 0.001
    assert stats.kstest(b, stats.norm(scale=np.sqrt(2.0)).cdf).pvalue > 0.001
    assert abs(np.corrcoef(a, b)[0, 1]) < 0.03          # independent streams
    eq1 = pde_hip.PDE({"a": "0*a", "b": "0*b"}, noise=0.5, rng=np.random.default_rng(4))
    res = eq1.solve(state, t_range=1, dt=1, solver="euler", backend="hip")
    assert stats.kstest(res.data.ravel(), stats.norm(scale=np.sqrt(0.5)).cdf).pvalue > 0.001
