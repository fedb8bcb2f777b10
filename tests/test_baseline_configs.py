"""Parity at the sizes BASELINE.json quotes (cfg1 .. cfg5), full field — VERDICT r1 row (g).

Two halves:

* ``-m "not gpu"``: the CPU oracle at full size against digests / samples produced by the REFERENCE itself
  (``tests/golden/configs.npz`` from ``tests/golden/make_golden_configs.py``; the reference's torch-CPU Euler runs and the
  oracle agree bit for bit, so a SHA-256 of the final state pins the oracle over the whole field).
* ``-m gpu``: the HIP library against the oracle over the WHOLE field (bit for bit in fp64, 1e-5 relative with equal step
  counts for the fp32 RKF45 configuration), through the mirror API (``eq.solve(..., backend="hip")``) — i.e. through
  ``pdehip_euler_run`` incl. its hipGraph replay for the launch-bound 2-D grids and the two-steps-per-sweep kernels.

cfg1 (UnitGrid 64^2) is the golden case ``diff64_euler_torch`` of ``steppers.npz`` (full field).
"""

from __future__ import annotations

import hashlib
import json
import time

import numpy as np
import pytest
from helpers import GOLDEN, host_faces, interior, max_rel, oracle_grid, to_full
from test_oracle_golden import oracle_solve

import pde_hip
from oracle import pde_oracle as O
from pde_hip import _abi

CONFIGS = np.load(GOLDEN / "configs.npz", allow_pickle=False)
CASES = {c["id"]: c for c in json.loads(str(CONFIGS["cases"]))}


def _initial(case) -> np.ndarray:
    """The reference's `ScalarField.random_uniform(grid, vmin, vmax, rng=default_rng(0))` (checked by the generator)."""
    return np.random.default_rng(0).uniform(case["vmin"], case["vmax"], size=case["shape"]).astype(case.get("dtype", "float64"))


def _grid(case):
    return pde_hip.CartesianGrid(case["bounds"], case["shape"], periodic=case["periodic"])


def _oracle_final(case):
    grid = _grid(case)
    dtype = np.dtype(case.get("dtype", "float64"))
    ocase = dict(case)
    if case["pde"] == "expression":
        ocase.update(pde="cahn_hilliard", gamma=1.0)
    if case.get("adaptive"):
        ocase["dt"] = None          # oracle_solve: dt None = adaptive loop starting at dt = 1e-3
    return oracle_solve(ocase, grid, dtype, _initial(case))


def _sample(case, data):
    return data[(slice(None, None, case["stride"]),) * len(case["shape"])]


# --------------------------------------------------------------------------------------------------------------
# oracle vs the reference at full size (CPU)
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cid", ["cfg2_diffusion_1024sq", "cfg3_cahn_hilliard_512sq", "cfg3_cahn_hilliard_512sq_10k"])
def test_oracle_matches_reference_digest(cid):
    case = CASES[cid]
    final, steps, _ = _oracle_final(case)
    assert steps == int(CONFIGS[f"{cid}/steps"])
    np.testing.assert_array_equal(_sample(case, final), CONFIGS[f"{cid}/sample"])
    assert hashlib.sha256(np.ascontiguousarray(final).tobytes()).hexdigest() == str(CONFIGS[f"{cid}/sha256"])


# --------------------------------------------------------------------------------------------------------------
# HIP vs oracle (and vs the reference digests) at full size (GPU)
# --------------------------------------------------------------------------------------------------------------
def _make_eq(case):
    if case["pde"] == "diffusion":
        return pde_hip.DiffusionPDE(case["D"], bc=case["bc"])
    if case["pde"] == "cahn_hilliard":
        return pde_hip.CahnHilliardPDE(case["gamma"], bc_c=case["bc"], bc_mu=case["bc"])
    return pde_hip.PDE(case["rhs"], bc=case["bc"])


@pytest.mark.gpu
def test_cfg1_unitgrid_64sq_euler(golden_steppers):
    """cfg1: DiffusionPDE on UnitGrid([64, 64]) fp64, Euler dt = 0.1, t_range = 10 — the reference's torch-CPU result."""
    cid = "diff64_euler_torch"
    grid = pde_hip.UnitGrid([64, 64])
    state = pde_hip.ScalarField(grid, golden_steppers[f"{cid}/input"])
    res, info = pde_hip.DiffusionPDE().solve(state, t_range=10, dt=0.1, solver="euler", backend="hip", ret_info=True)
    assert info["solver"]["steps"] == 100
    np.testing.assert_array_equal(res.data, golden_steppers[f"{cid}/final"])


@pytest.mark.gpu
@pytest.mark.parametrize("cid", ["cfg2_diffusion_1024sq", "cfg3_cahn_hilliard_512sq"])
def test_cfg2_cfg3_full_field_1000_steps(cid):
    """cfg2: 1024^2 periodic diffusion; cfg3: UnitGrid 512^2 Cahn-Hilliard (the reference's published benchmark problem,
    scripts/performance_solvers.py:53-66) — 1000 Euler steps through pdehip_euler_run (hipGraph replay of the launch-bound
    loop, two-level kernels), whole field == the reference's torch-CPU run (digest) == the oracle, bit for bit."""
    case = CASES[cid]
    grid = _grid(case)
    state = pde_hip.ScalarField(grid, _initial(case))
    res, info = _make_eq(case).solve(state, t_range=case["t_range"], dt=case["dt"], solver="euler", backend="hip", ret_info=True)
    assert info["solver"]["steps"] == int(CONFIGS[f"{cid}/steps"]) == 1000
    np.testing.assert_array_equal(_sample(case, res.data), CONFIGS[f"{cid}/sample"])
    assert hashlib.sha256(np.ascontiguousarray(res.data).tobytes()).hexdigest() == str(CONFIGS[f"{cid}/sha256"])
    final, _, _ = _oracle_final(case)
    np.testing.assert_array_equal(res.data, final)


@pytest.mark.gpu
def test_cfg3_published_benchmark_10k_steps():
    """cfg3 at the step count of the reference's published benchmark (UnitGrid 512^2 Cahn-Hilliard, 10^4 Euler steps,
    scripts/performance_solvers.py:53-66, :140): the whole field equals the reference's torch-CPU run (digest) bit for bit."""
    cid = "cfg3_cahn_hilliard_512sq_10k"
    case = CASES[cid]
    state = pde_hip.ScalarField(_grid(case), _initial(case))
    t0 = time.time()
    res, info = _make_eq(case).solve(state, t_range=case["t_range"], dt=case["dt"], solver="euler", backend="hip", ret_info=True)
    t_hip = time.time() - t0
    assert info["solver"]["steps"] == int(CONFIGS[f"{cid}/steps"]) == 10000
    np.testing.assert_array_equal(_sample(case, res.data), CONFIGS[f"{cid}/sample"])
    assert hashlib.sha256(np.ascontiguousarray(res.data).tobytes()).hexdigest() == str(CONFIGS[f"{cid}/sha256"])
    print(f"cfg3 10^4 steps: hip {t_hip:.3f}s (reference: 43.7 s with numba on an M4 Pro, BASELINE.md)")


@pytest.mark.gpu
def test_cfg4_512cube_full_field():
    """cfg4 (512^3 fp64 periodic): 6 Euler steps (three two-step sweeps: the bench kernel, x-chunk seams included), one RK4
    step, one RKF45 attempt and one Cahn-Hilliard Euler step — every cell compared with the oracle run on the whole grid."""
    import ctypes as C

    from pde_hip.device import DeviceArray, DeviceScalar, ptr_array

    case = CASES["cfg4_diffusion_512cube_6steps"]
    grid = _grid(case)
    backend = pde_hip.get_backend("hip")
    lib = backend._lib
    u = _initial(case)
    g = oracle_grid(grid)
    faces = host_faces(grid.get_boundary_conditions("auto_periodic_neumann"))
    full = to_full(grid, u)
    # --- 6 Euler steps: HIP == oracle == the reference's torch-CPU digest
    state = pde_hip.ScalarField(grid, u)
    res = pde_hip.DiffusionPDE().solve(state, t_range=0.6, dt=0.1, solver="euler", backend="hip")
    rhs = O.make_rhs(_abi.RHS_DIFFUSION, 1.0, faces.c)
    expect = interior(grid, O.euler_run(g, rhs, full, 0.1, 6))
    np.testing.assert_array_equal(res.data, expect)
    if "cfg4_diffusion_512cube_6steps/sha256" in CONFIGS.files:
        assert hashlib.sha256(np.ascontiguousarray(res.data).tobytes()).hexdigest() == str(CONFIGS["cfg4_diffusion_512cube_6steps/sha256"])
    del res, expect
    # --- one RK4 step and one RKF45 attempt of the diffusion equation (stage-fused sweeps)
    eq = pde_hip.DiffusionPDE()
    spec = backend.make_rhs_spec(eq, state)
    y = DeviceArray(spec.info).set_valid(u)
    work = [DeviceArray(spec.info) for _ in range(7)]
    lib.rk4_step(spec.info.ref, spec.ref, y.ptr, ptr_array(work[:5]), 0.05, None)
    o = full.copy()
    O.rk4_step(g, rhs, o, 0.05)
    np.testing.assert_array_equal(y.get_valid(), interior(grid, o))
    del o
    y.set_valid(u)
    ynew, err = DeviceArray(spec.info), DeviceScalar()
    lib.rkf45_attempt(spec.info.ref, spec.ref, y.ptr, ynew.ptr, ptr_array(work), 0.05, err.ptr, None)
    onew, oerr = O.rkf45_attempt(g, rhs, full.copy(), 0.05)
    np.testing.assert_array_equal(ynew.get_valid(), interior(grid, onew))
    assert err.value() == oerr
    del onew, work, ynew, y
    # --- one Cahn-Hilliard Euler step (fused two-level sweep, mu in registers)
    state_ch = pde_hip.ScalarField(grid, u - 0.5)
    res = pde_hip.CahnHilliardPDE().solve(state_ch, t_range=1e-3, dt=1e-3, solver="euler", backend="hip")
    scratch = np.zeros_like(full)
    rhs_ch = O.make_rhs(_abi.RHS_CAHN_HILLIARD, 1.0, faces.c, faces.c, scratch)
    expect = interior(grid, O.euler_run(g, rhs_ch, to_full(grid, u - 0.5), 1e-3, 1))
    np.testing.assert_array_equal(res.data, expect)


@pytest.mark.gpu
@pytest.mark.parametrize("periodic", [False, [True, False, False]], ids=["walls", "periodic-march-axis"])
def test_fields_beyond_400MB_with_faces(periodic):
    """fp64 fields beyond 400 MB with local faces (the streaming-store form of the 4-row tile, its face code in branches): 4 Euler steps (two
    sweeps, the x-chunk seams included) of 224 x 512 x 512 with value / derivative / mixed conditions, every cell against the oracle."""
    grid = pde_hip.CartesianGrid([[0, 2], [0, 3], [0, 1]], (224, 512, 512), periodic=periodic)
    bc = {"x": "periodic"} if periodic else {"x-": {"value": 0.2}, "x+": {"derivative": 0.1}}
    bc.update({"y-": {"value": -0.1}, "y+": {"derivative": 0.3}, "z-": {"type": "mixed", "value": 0.7, "const": 0.2}, "z+": {"value": 0.4}})
    rng = np.random.default_rng(5)
    u = rng.uniform(-1, 1, grid.shape)
    D, dt = 0.8, 0.1 * float(min(grid.discretization)) ** 2
    res = pde_hip.DiffusionPDE(D, bc=bc).solve(pde_hip.ScalarField(grid, u), t_range=4 * dt, dt=dt, solver="euler", backend="hip", tracker=None)
    name = pde_hip.get_backend("hip")._lib.last_kernel_name().decode()
    g = oracle_grid(grid)
    faces = host_faces(grid.get_boundary_conditions(bc))
    rhs = O.make_rhs(_abi.RHS_DIFFUSION, D, faces.c)
    expect = interior(grid, O.euler_run(g, rhs, to_full(grid, u), dt, 4))
    np.testing.assert_array_equal(res.data, expect)
    assert np.abs(res.data - u).max() > 1e-3
    assert "euler2_kernel<double,2,4" in name and "NT" in name, name


@pytest.mark.gpu
@pytest.mark.parametrize("cid", ["cfg5_expression_256cube_f32_rkf45", "cfg5_expression_256cube_f32_rkf45_long"])
def test_cfg5_256cube_f32_expression_rkf45(cid):
    """cfg5: PDE({'c': 'laplace(c**3 - c - laplace(c))'}) on 256^3 fp32, adaptive RKF45 (tolerance 1e-4, dt0 = 1e-3):
    equal step count and <= 1e-5 relative against the oracle's adaptive loop over the whole field, and against the
    reference's numpy+scipy run (sample) — fp32 contract of include/pdehip.h (fp32 storage, fp64 registers).
    `_long`: t_range = 2.6, i.e. >= 100 accepted steps."""
    if cid not in CASES or f"{cid}/steps" not in CONFIGS.files:
        pytest.skip(f"golden data of {cid} not generated")
    case = CASES[cid]
    grid = _grid(case)
    u = _initial(case)
    state = pde_hip.ScalarField(grid, u, dtype=np.float32)
    t0 = time.time()
    res, info = _make_eq(case).solve(state, t_range=case["t_range"], dt=case["dt"], solver="runge-kutta", adaptive=True, backend="hip",
                                      ret_info=True)
    t_hip = time.time() - t0
    final, steps, dt_last = _oracle_final(case)
    assert res.data.dtype == np.float32
    assert info["solver"]["steps"] == steps
    np.testing.assert_allclose(info["solver"]["dt"], dt_last, rtol=1e-4)
    assert max_rel(res.data.astype(np.float64), final.astype(np.float64)) < 1e-5
    if f"{cid}/sample" in CONFIGS.files:
        assert info["solver"]["steps"] == int(CONFIGS[f"{cid}/steps"])
        ref = CONFIGS[f"{cid}/sample"].astype(np.float64)
        err_ref = max_rel(_sample(case, res.data).astype(np.float64), ref)
        if not cid.endswith("_long"):
            assert err_ref < 1e-5
        else:
            # >= 100 accepted steps: the reference's run is PURE fp32 (numpy + scipy), this one rounds to fp32 once per stored
            # value (fp32 storage, fp64 registers): two different rounding sequences drift apart as the steps add up (1.5e-5
            # after 69 steps).  What can be asserted is that this run is the one closer to the exact (fp64) trajectory: the
            # same adaptive run of the oracle in fp64 is the yardstick for both.
            assert steps >= 100
            assert err_ref < 1e-4
            truth, steps64, _ = _oracle_final({**case, "dtype": "float64"})
            err_hip_truth = max_rel(res.data.astype(np.float64), truth)
            err_ref_truth = max_rel(ref, _sample(case, truth))
            print(f"cfg5 long: vs fp64 run: hip {err_hip_truth:.2e}, reference (pure fp32) {err_ref_truth:.2e}; hip vs reference {err_ref:.2e}; fp64 steps {steps64}")
            assert err_hip_truth < 2e-5 and err_hip_truth <= 1.5 * err_ref_truth + 1e-6
    print(f"cfg5: {steps} accepted steps, hip {t_hip:.2f}s")
