"""Temporal-blocking kernel (two Euler steps per sweep, pdehip_march2.inc) == two single steps of the oracle.

Bit-exact for fp64 and fp32: the intermediate level is rounded to the storage type and the BCs are applied
to it exactly like the reference does between two steps (pde/backends/numba/_solvers.py:98-108 loop body
with pde/solvers/euler.py:172-175).  Geometry cases cover one / several tiles along every axis, several
x-chunks (recomputed halo planes) and every combination of periodic / local faces per axis.
"""

from __future__ import annotations

import ctypes as C
import itertools

import numpy as np
import pytest
from helpers import host_faces, interior, oracle_grid, to_full

import pde_hip
from oracle import pde_oracle as O
from pde_hip import _abi
from pde_hip.backend import convert_bcs
from pde_hip.device import DeviceArray, GridInfo

pytestmark = pytest.mark.gpu

LOCAL = [
    ({"value": 0.5}, {"derivative": 0.25}),
    ({"type": "mixed", "value": 1.5, "const": 0.2}, {"value": -0.3}),
    ({"derivative": -1.0}, {"type": "mixed", "value": -0.5, "const": 1.0}),
]


def _setup(shape, periodic, dtype, seed=3):
    grid = pde_hip.CartesianGrid([[0, n * (0.7 + 0.1 * i)] for i, n in enumerate(shape)], shape, periodic=periodic)
    bc = {}
    for i, a in enumerate(grid.axes):
        if periodic[i]:
            bc[a] = "periodic"
        else:
            bc[a + "-"], bc[a + "+"] = LOCAL[i]
    bcs = grid.get_boundary_conditions(bc)
    data = np.random.default_rng(seed).uniform(-0.5, 0.5, shape).astype(dtype)
    return grid, bc, bcs, data


def _oracle_steps(grid, bcs, data, D, dt, steps):
    g = oracle_grid(grid, data.dtype)
    rhs = O.make_rhs(_abi.RHS_DIFFUSION, D, host_faces(bcs).c)
    return interior(grid, O.euler_run(g, rhs, to_full(grid, data), dt, steps))


@pytest.fixture(scope="module")
def backend():
    return pde_hip.get_backend("hip")


def _euler2(backend, grid, bcs, data, D, dt):
    info = GridInfo(grid.shape, grid.discretization, data.dtype)
    faces = convert_bcs(bcs)
    a, b = DeviceArray(info).set_valid(data), DeviceArray(info)
    done = C.c_int(0)
    backend._lib.diffusion_euler2(info.ref, faces.c, a.ptr, b.ptr, D, dt, C.byref(done), None)
    return done.value, b.get_valid()


PERIODIC = list(itertools.product([True, False], repeat=3))


@pytest.mark.parametrize("periodic", PERIODIC, ids=["".join("P" if p else "L" for p in per) for per in PERIODIC])
@pytest.mark.parametrize("dtype,shape", [
    (np.float64, (8, 8, 128)),      # one tile per plane, one x-chunk
    (np.float64, (40, 12, 256)),    # 2 chunks along z, 3 tiles along y, several x-chunks
    (np.float64, (5, 6, 128)),      # 2-row tiles (6 % 4 != 0), odd plane count
    (np.float32, (9, 8, 256)),
    (np.float32, (33, 16, 512)),
    (np.float64, (6, 8, 64)),       # row shorter than a chunk
    (np.float64, (7, 4, 200)),      # row ends inside the second chunk (36 of 64 lanes own cells)
    (np.float64, (5, 4, 130)),      # ... one lane of the last chunk owns cells
    (np.float32, (6, 8, 100)),
    # overlapped last tiles: rows that end inside a vector, row counts that no tile divides
    (np.float64, (9, 7, 129)),      # ONE cell in the last chunk: the chunk is moved back by one cell; 7 rows: 2-row tiles, the last one moved back
    (np.float64, (6, 37, 255)),     # the moved chunk is a full one; 37 rows: 4-row tiles, the last one moved back by 3 rows
    (np.float64, (5, 9, 383)),
    (np.float32, (6, 5, 259)),      # wide fp32 tile, 3 cells beyond the first chunk
    (np.float32, (7, 9, 257)),
    (np.float32, (4, 35, 518)),
    (np.float32, (5, 6, 130)),      # row shorter than the wide chunk: the narrow tile takes it
    # "open" rows (one or two cells beyond whole chunks): the tiles cover the chunks, pdehip_shell.hip the remaining columns
    (np.float64, (12, 20, 257)),
    (np.float64, (10, 8, 258)),
    (np.float32, (9, 16, 513)),     # wide fp32 tile when the fastest axis is periodic, the narrow one (virtual far column) otherwise
    (np.float32, (6, 10, 514)),
    (np.float32, (6, 8, 385)),      # 3 x 128 + 1: the narrow tile
    (np.float64, (6, 8, 131)),      # three and four columns: two jobs of the shell kernel
    (np.float64, (5, 12, 260)),
    (np.float32, (7, 8, 515)),
    (np.float32, (6, 6, 388)),
    (np.float32, (5, 8, 263)),      # seven and eight columns: four jobs
    (np.float64, (7, 9, 135)),
    (np.float32, (6, 10, 520)),
])
def test_two_steps_per_sweep_equal_two_single_steps(backend, periodic, dtype, shape):
    grid, bc, bcs, data = _setup(shape, list(periodic), dtype)
    done, got = _euler2(backend, grid, bcs, data, 0.6, 2e-3)
    assert done == 1
    np.testing.assert_array_equal(got, _oracle_steps(grid, bcs, data, 0.6, 2e-3, 2))


@pytest.mark.parametrize("steps", [1, 2, 5, 8])
@pytest.mark.parametrize("periodic", [(True, True, True), (False, True, False)])
def test_euler_run_uses_pairs_and_stays_bit_exact(backend, steps, periodic):
    grid, bc, bcs, data = _setup((12, 8, 128), list(periodic), np.float64, seed=5)
    eq = pde_hip.DiffusionPDE(0.8, bc=bc)
    state = pde_hip.ScalarField(grid, data)
    spec = backend.make_rhs_spec(eq, state)
    a, b = DeviceArray(spec.info).set_valid(data), DeviceArray(spec.info)
    res = C.c_void_p()
    backend._lib.euler_run(spec.info.ref, spec.ref, a.ptr, b.ptr, 1e-3, steps, C.byref(res), None)
    got = (b if res.value == b.ptr else a).get_valid()
    np.testing.assert_array_equal(got, _oracle_steps(grid, bcs, data, 0.8, 1e-3, steps))
    # an odd number of sweeps leaves the result in the second buffer (1 step: 1 sweep, 2: 1, 5: 3, 8: 4 sweeps)
    sweeps = steps // 2 + steps % 2
    assert (res.value == b.ptr) == (sweeps % 2 == 1)


def test_cases_outside_the_kernel_report_not_done(backend):
    for shape, periodic, bc_override in [
        ((8, 8, 63), [True] * 3, None),                       # odd row length inside ONE chunk: no neighbour to overlap with
        ((8, 8, 128), [False] * 3, {"curvature": 0.3}),       # second-order faces
        ((8, 8, 128), [True] * 3, "anti-periodic"),           # wraps with a factor -1
    ]:
        grid, bc, bcs, data = _setup(shape, periodic, np.float64)
        if bc_override is not None:
            bcs = grid.get_boundary_conditions(bc_override)
        done, _ = _euler2(backend, grid, bcs, data, 0.6, 2e-3)
        assert done == 0
    # ... and the time loop silently takes single steps there (same results as before)
    grid, bc, bcs, data = _setup((6, 7, 63), [True, False, True], np.float64)
    eq = pde_hip.DiffusionPDE(0.8, bc=bc)
    spec = backend.make_rhs_spec(eq, pde_hip.ScalarField(grid, data))
    a, b = DeviceArray(spec.info).set_valid(data), DeviceArray(spec.info)
    res = C.c_void_p()
    backend._lib.euler_run(spec.info.ref, spec.ref, a.ptr, b.ptr, 1e-3, 4, C.byref(res), None)
    np.testing.assert_array_equal((b if res.value == b.ptr else a).get_valid(), _oracle_steps(grid, bcs, data, 0.8, 1e-3, 4))


LOCAL_MU = [
    ({"derivative": 0.3}, {"value": 0.1}),
    ({"value": -0.2}, {"type": "mixed", "value": 0.7, "const": -0.4}),
    ({"type": "mixed", "value": -1.5, "const": 0.3}, {"derivative": 0.6}),
]


@pytest.mark.parametrize("periodic", PERIODIC, ids=["".join("P" if p else "L" for p in per) for per in PERIODIC])
@pytest.mark.parametrize("dtype,shape", [
    (np.float64, (8, 8, 128)),
    (np.float64, (37, 12, 256)),
    (np.float64, (5, 6, 128)),
    (np.float32, (9, 8, 256)),
    (np.float64, (6, 4, 72)),
    (np.float32, (5, 4, 264)),
    (np.float64, (7, 9, 131)),      # overlapped last tiles along the rows and along the row
    (np.float64, (5, 34, 255)),
    (np.float32, (5, 7, 261)),
])
def test_cahn_hilliard_in_one_sweep_equals_two_kernels(backend, periodic, dtype, shape):
    """mu = c^3 - c - g lap(c) (faces of c) and lap(mu) (DIFFERENT faces of mu) fused with mu in registers ==
    oracle Euler step / oracle dt*rhs, bit-exact."""
    grid, bc_c, bcs_c, data = _setup(shape, list(periodic), dtype, seed=9)
    bc_mu = {}
    for i, a in enumerate(grid.axes):
        if periodic[i]:
            bc_mu[a] = "periodic"
        else:
            bc_mu[a + "-"], bc_mu[a + "+"] = LOCAL_MU[i]
    bcs_mu = grid.get_boundary_conditions(bc_mu)
    info = GridInfo(grid.shape, grid.discretization, data.dtype)
    fc, fm = convert_bcs(bcs_c), convert_bcs(bcs_mu)
    g = oracle_grid(grid, dtype)
    scratch = np.zeros(grid._shape_full, dtype)
    orhs = O.make_rhs(_abi.RHS_CAHN_HILLIARD, 0.8, host_faces(bcs_c).c, host_faces(bcs_mu).c, scratch)
    a, b = DeviceArray(info).set_valid(data), DeviceArray(info)
    done = C.c_int(0)
    backend._lib.cahn_hilliard_fused(info.ref, fc.c, fm.c, a.ptr, b.ptr, 0.8, 1e-3, 1, C.byref(done), None)
    assert done.value == 1
    np.testing.assert_array_equal(b.get_valid(), interior(grid, O.euler_run(g, orhs, to_full(grid, data), 1e-3, 1)))
    backend._lib.cahn_hilliard_fused(info.ref, fc.c, fm.c, a.ptr, b.ptr, 0.8, 0.02, 0, C.byref(done), None)
    assert done.value == 1
    np.testing.assert_array_equal(b.get_valid(), interior(grid, O.rhs_scaled(g, orhs, to_full(grid, data), 0.02)))


def test_cahn_hilliard_solvers_use_the_fused_sweep(backend):
    """Euler / RK4 / RKF45 through the stepper entry points on a covered grid (the fused sweep is taken inside)."""
    grid, bc, bcs, data = _setup((10, 8, 128), [True, False, False], np.float64, seed=12)
    eq = pde_hip.CahnHilliardPDE(0.9, bc_c=bc, bc_mu=bc)
    state = pde_hip.ScalarField(grid, data)
    from test_oracle_golden import oracle_solve

    for solver, dt in [("euler", 1e-3), ("runge-kutta", 1e-3), ("runge-kutta", None)]:
        res, info = eq.solve(state, t_range=0.01, dt=dt, solver=solver, backend="hip", ret_info=True)
        case = {"bc": bc, "t_range": 0.01, "dt": dt, "solver": solver, "pde": "cahn_hilliard", "gamma": 0.9}
        expect, steps, _ = oracle_solve(case, grid, np.float64, data)
        assert info["solver"]["steps"] == steps
        np.testing.assert_array_equal(res.data, expect)
    # faces of c and mu periodic on different axes cannot be fused: two kernels, same numbers as the oracle
    grid2 = pde_hip.UnitGrid([8, 8, 128], periodic=[True, True, False])
    info2 = GridInfo(grid2.shape, grid2.discretization, np.float64)
    f1 = convert_bcs(grid2.get_boundary_conditions("auto_periodic_neumann"))
    f2 = convert_bcs(grid2.get_boundary_conditions({"x": "periodic", "y": "anti-periodic", "z": {"value": 0}}))
    a, b = DeviceArray(info2).set_valid(np.zeros(grid2.shape)), DeviceArray(info2)
    done = C.c_int(1)
    backend._lib.cahn_hilliard_fused(info2.ref, f1.c, f2.c, a.ptr, b.ptr, 1.0, 1e-3, 1, C.byref(done), None)
    assert done.value == 0


PERIODIC2 = list(itertools.product([True, False], repeat=2))


@pytest.mark.parametrize("periodic", PERIODIC2, ids=["".join("P" if p else "L" for p in per) for per in PERIODIC2])
@pytest.mark.parametrize("dtype,shape", [
    (np.float64, (16, 128)),
    (np.float64, (41, 256)),     # several chunks along the march axis (chunks of >= 4 rows), odd row count
    (np.float64, (9, 200)),      # row ends inside the second chunk
    (np.float64, (64, 64)),      # BASELINE config 1 shape
    (np.float32, (12, 256)),
    (np.float32, (7, 100)),
    (np.float64, (11, 131)),     # the last chunk moved back by one cell
    (np.float32, (9, 262)),
    (np.float64, (11, 129)),     # open rows in 2-D
    (np.float32, (9, 258)),
    (np.float32, (12, 514)),     # (one cell beyond the chunks with a local face there: the wide fp32 tile - the only 2-D fp32 one - declines)
])
def test_two_dimensional_grids(backend, periodic, dtype, shape):
    """2-D: the same two-level kernel marching along the first axis (no rows): two diffusion steps per sweep and the
    Cahn-Hilliard right-hand side in one sweep, bit-exact against the oracle."""
    grid, bc, bcs, data = _setup(shape, list(periodic), dtype, seed=21)
    done, got = _euler2(backend, grid, bcs, data, 0.6, 2e-3)
    assert done == 1
    np.testing.assert_array_equal(got, _oracle_steps(grid, bcs, data, 0.6, 2e-3, 2))
    bc_mu = {}
    for i, a in enumerate(grid.axes):
        if periodic[i]:
            bc_mu[a] = "periodic"
        else:
            bc_mu[a + "-"], bc_mu[a + "+"] = LOCAL_MU[i]
    bcs_mu = grid.get_boundary_conditions(bc_mu)
    info = GridInfo(grid.shape, grid.discretization, data.dtype)
    fc, fm = convert_bcs(bcs), convert_bcs(bcs_mu)
    g = oracle_grid(grid, dtype)
    scratch = np.zeros(grid._shape_full, dtype)
    orhs = O.make_rhs(_abi.RHS_CAHN_HILLIARD, 0.8, host_faces(bcs).c, host_faces(bcs_mu).c, scratch)
    a, b = DeviceArray(info).set_valid(data), DeviceArray(info)
    d = C.c_int(0)
    backend._lib.cahn_hilliard_fused(info.ref, fc.c, fm.c, a.ptr, b.ptr, 0.8, 1e-3, 1, C.byref(d), None)
    assert d.value == 1
    np.testing.assert_array_equal(b.get_valid(), interior(grid, O.euler_run(g, orhs, to_full(grid, data), 1e-3, 1)))
    backend._lib.cahn_hilliard_fused(info.ref, fc.c, fm.c, a.ptr, b.ptr, 0.8, 0.02, 0, C.byref(d), None)
    assert d.value == 1
    np.testing.assert_array_equal(b.get_valid(), interior(grid, O.rhs_scaled(g, orhs, to_full(grid, data), 0.02)))
    # the stepper entry point: 9 steps = 4 double sweeps + 1 single step
    eq = pde_hip.DiffusionPDE(0.8, bc=bc)
    spec = backend.make_rhs_spec(eq, pde_hip.ScalarField(grid, data, dtype=dtype))
    a, b = DeviceArray(spec.info).set_valid(data), DeviceArray(spec.info)
    res = C.c_void_p()
    backend._lib.euler_run(spec.info.ref, spec.ref, a.ptr, b.ptr, 1e-3, 9, C.byref(res), None)
    np.testing.assert_array_equal((b if res.value == b.ptr else a).get_valid(), _oracle_steps(grid, bcs, data, 0.8, 1e-3, 9))


def test_full_size_time_loop_512cubed(backend):
    """BASELINE size (512^3 fp64, periodic, Euler dt=0.1): 6 steps of pdehip_euler_run (3 double sweeps).

    * slabs of the result equal the oracle run on those slabs alone with 6 spare layers per side (an error at the slab
      end travels one layer per step, so the inner layers are exact) - bit for bit,
    * the periodic sum of the field is conserved by every diffusion step (to rounding),
    * the maximum principle: the range of the field shrinks.
    """
    n, steps, dt = 512, 6, 0.1
    grid = pde_hip.UnitGrid([n, n, n], periodic=True)
    u = np.random.default_rng(0).random((n, n, n))
    eq = pde_hip.DiffusionPDE(1.0)
    spec = backend.make_rhs_spec(eq, pde_hip.ScalarField(grid, u))
    a, b = DeviceArray(spec.info).set_valid(u), DeviceArray(spec.info)
    done = C.c_int(0)
    backend._lib.diffusion_euler2(spec.info.ref, spec.bc_c.c, a.ptr, b.ptr, 1.0, dt, C.byref(done), None)
    assert done.value == 1          # the two-level kernel covers the benchmark grid
    a.set_valid(u)
    res = C.c_void_p()
    backend._lib.euler_run(spec.info.ref, spec.ref, a.ptr, b.ptr, dt, steps, C.byref(res), None)
    got = (b if res.value == b.ptr else a).get_valid()
    pad, keep = steps, 4
    for lo in (0, 251, n - keep):
        idx = np.arange(lo - pad, lo + keep + pad) % n
        sub = pde_hip.CartesianGrid([[0, len(idx)], [0, n], [0, n]], [len(idx), n, n], periodic=[False, True, True])
        bcs = sub.get_boundary_conditions({"x": {"derivative": 0}, "y": "periodic", "z": "periodic"})
        rhs = O.make_rhs(_abi.RHS_DIFFUSION, 1.0, host_faces(bcs).c)
        out = interior(sub, O.euler_run(oracle_grid(sub), rhs, to_full(sub, u[idx]), dt, steps))
        np.testing.assert_array_equal(got[lo:lo + keep], out[pad:pad + keep])
    assert abs(got.sum() - u.sum()) < 1e-9 * u.sum()
    assert got.min() > u.min() and got.max() < u.max()


@pytest.mark.parametrize("shape,periodic", [((9, 8, 128), [True, False, True]), ((6, 4, 72), [False, True, False]), ((20, 136), [False, True]),
                                            ((7, 9, 131), [False, False, False])])
def test_special_values_stay_bit_identical(backend, shape, periodic):
    """Denormals, huge magnitudes, signed zeros, an infinity and a NaN travel through both levels exactly as through two
    single steps of the oracle (IEEE arithmetic, denormals on, no contraction, selects instead of arithmetic masking)."""
    grid, bc, bcs, data = _setup(shape, periodic, np.float64, seed=31)
    rng = np.random.default_rng(32)
    data = data * 10.0 ** rng.integers(-320, 300, size=shape).astype(np.float64)
    flat = data.reshape(-1)
    flat[rng.choice(flat.size, 40, replace=False)] = 0.0
    flat[rng.choice(flat.size, 40, replace=False)] = -0.0
    flat[rng.choice(flat.size, 20, replace=False)] = 4.9e-324
    flat[flat.size // 3] = np.inf
    flat[2 * flat.size // 3] = np.nan
    with np.errstate(all="ignore"):
        done, got = _euler2(backend, grid, bcs, data, 0.6, 2e-3)
        expect = _oracle_steps(grid, bcs, data, 0.6, 2e-3, 2)
    assert done == 1
    np.testing.assert_array_equal(got, expect)
    assert np.array_equal(np.signbit(got[got == 0]), np.signbit(expect[expect == 0]))   # signed zeros too
    assert np.isnan(got).sum() > 1 and np.isinf(got).sum() >= 0


@pytest.mark.parametrize("dtype,shape,split", [(np.float64, (12, 8, 128), 5), (np.float64, (9, 4, 72), 4), (np.float32, (10, 6, 256), 6),
                                               (np.float64, (11, 9, 131), 5), (np.float32, (9, 7, 259), 4)])
def test_sub_slabs_with_one_physical_face(backend, dtype, shape, split):
    """First / last slab of a NON-periodic slowest axis: one side of the sub-slab is the physical (local) face, the other
    side reads two real layers of the neighbouring sub-slab.  Two such launches over the halves of one array must equal
    the sweep over the whole array (and the oracle) bit for bit - this is what the first and the last rank of
    pdehip_slab_euler2_run execute."""
    grid, bc, bcs, data = _setup(shape, [False, False, True], dtype, seed=41)
    info = GridInfo(grid.shape, grid.discretization, data.dtype)
    faces = convert_bcs(bcs)
    a, whole, parts = DeviceArray(info).set_valid(data), DeviceArray(info), DeviceArray(info)
    done = C.c_int(0)
    lib = backend._lib
    lib.diffusion_euler2(info.ref, faces.c, a.ptr, whole.ptr, 0.6, 2e-3, C.byref(done), None)
    assert done.value == 1
    n0 = shape[0]
    lp = info.layer_pitch * data.dtype.itemsize
    for first, count, sides in [(1, split, 2), (split + 1, n0 - split, 3)]:
        sub = GridInfo((count, *shape[1:]), grid.discretization, data.dtype)
        f = _abi.FaceArray()
        for i in range(6):
            f[i] = faces.c[i]
        f[0 if sides == 3 else 1].kind = _abi.BC_SKIP               # the side with real layers
        f[1].index1 -= first - 1                                    # upper face index relative to the sub-slab
        lib.diffusion_euler2_slab(sub.ref, f, a.ptr + (first - 1) * lp, parts.ptr + (first - 1) * lp, 0.6, 2e-3, sides, C.byref(done), None)
        assert done.value == 1
    np.testing.assert_array_equal(parts.get_valid(), whole.get_valid())
    np.testing.assert_array_equal(whole.get_valid(), _oracle_steps(grid, bcs, data, 0.6, 2e-3, 2))


@pytest.mark.parametrize("tile", ["default", "4,2,4,1", "2,4,2,4", "2,2,2,2", "2,1,2,1", "4,1,4,1"])
def test_fp32_tile_shapes(tile):
    """Every fp32 wave tile of the two-level kernel (wide: 4 cells per lane, 2 or 1 rows; narrow: 2 cells per lane, 4 / 2 / 1
    rows; the stage epilogue of the Runge-Kutta sweeps on each) gives the oracle's bits: Euler runs, RK4 step, RKF45 attempt
    (tests/f32_tile_worker.py in a process of its own - the tile choice is read once per process)."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    env = dict(os.environ)
    env.pop("PDEHIP_F32_TILE", None)
    if tile != "default":
        env["PDEHIP_F32_TILE"] = tile
    worker = Path(__file__).resolve().parent / "f32_tile_worker.py"
    proc = subprocess.run([sys.executable, str(worker)], capture_output=True, text=True, timeout=600, env=env)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("F32TILE ")]
    assert proc.returncode == 0 and lines and lines[-1] == "F32TILE OK", (proc.stdout[-2000:], proc.stderr[-2000:])
