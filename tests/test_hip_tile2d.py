"""K Euler steps of a 2-D grid per launch, time levels in LDS (csrc/pdehip_tile2d.inc) — VERDICT r1 item 10 (BASELINE
configs 1-3 are launch-bound).  Bit-identical to K single steps of the oracle for every periodic / local face
combination, tile-boundary and grid-smaller-than-tile geometries, fp64 and fp32, diffusion and Cahn-Hilliard; and
`pdehip_euler_run` takes it by itself (remainders, hipGraph replay).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import pytest
from helpers import host_faces, interior, oracle_grid, to_full

import pde_hip
from oracle import pde_oracle as O
from pde_hip import _abi
from pde_hip.device import DeviceArray

pytestmark = pytest.mark.gpu

LOCAL_X = {"x-": {"value": 0.4}, "x+": {"derivative": -0.2}}
LOCAL_Y = {"y-": {"type": "mixed", "value": 0.5, "const": 0.2}, "y+": {"value": -0.3}}
FACES = {
    "pp": ([True, True], "auto_periodic_neumann"),
    "pl": ([True, False], {"x": "periodic", **LOCAL_Y}),
    "lp": ([False, True], {**LOCAL_X, "y": "periodic"}),
    "ll": ([False, False], {**LOCAL_X, **LOCAL_Y}),
    "nn": ([False, False], "auto_periodic_neumann"),
}


def _oracle(grid, bc_c, bc_mu, kind, param, data, dt, steps, dtype):
    g = oracle_grid(grid, dtype)
    scratch = np.zeros(grid._shape_full, dtype)
    rhs = O.make_rhs(kind, param, host_faces(grid.get_boundary_conditions(bc_c)).c, host_faces(grid.get_boundary_conditions(bc_mu)).c, scratch)
    return interior(grid, O.euler_run(g, rhs, to_full(grid, data), dt, steps))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind", ["diffusion", "cahn_hilliard"])
@pytest.mark.parametrize("faces", list(FACES))
@pytest.mark.parametrize("shape", [(64, 64), (33, 70), (5, 7), (1, 9), (9, 1), (100, 130), (32, 64), (31, 129)])
def test_multi_step_launch_equals_single_steps(shape, faces, kind, dtype):
    periodic, bc = FACES[faces]
    grid = pde_hip.CartesianGrid([[0, 1.0 * shape[0]], [0, 0.9 * shape[1]]], shape, periodic=periodic)
    backend = pde_hip.get_backend("hip")
    data = np.random.default_rng(3).uniform(-0.4, 0.4, shape).astype(dtype)
    if kind == "diffusion":
        eq, code, param, dt, kmax = pde_hip.DiffusionPDE(0.7, bc=bc), _abi.RHS_DIFFUSION, 0.7, 0.05, 8
    else:
        eq, code, param, dt, kmax = pde_hip.CahnHilliardPDE(0.9, bc_c=bc, bc_mu=bc), _abi.RHS_CAHN_HILLIARD, 0.9, 1e-3, 4
    spec = backend.make_rhs_spec(eq, pde_hip.ScalarField(grid, data, dtype=dtype))
    a, b = DeviceArray(spec.info), DeviceArray(spec.info)
    for k in sorted({1, 2, 3, kmax}):
        a.set_valid(data)
        done = C.c_int(0)
        backend._lib.euler_multi_2d(spec.info.ref, spec.ref, a.ptr, b.ptr, dt, k, C.byref(done), None)
        assert done.value == 1
        np.testing.assert_array_equal(b.get_valid(), _oracle(grid, bc, bc, code, param, data, dt, k, dtype), err_msg=f"k={k}")
        np.testing.assert_array_equal(a.get_valid(), data)            # the input is not written
    backend._lib.euler_multi_2d(spec.info.ref, spec.ref, a.ptr, b.ptr, dt, kmax + 1, C.byref(done), None)
    assert done.value == 0                                            # more steps than the halo carries


def test_not_covered_cases_report_done_0():
    backend = pde_hip.get_backend("hip")
    done = C.c_int(1)
    for grid, bc in [(pde_hip.UnitGrid([16, 16], periodic=[False, True]), {"x": {"value": np.linspace(0, 1, 16)}, "y": "periodic"}),     # per-face array
                     (pde_hip.UnitGrid([16, 16], periodic=[False, True]), {"x": "extrapolate", "y": "periodic"}),                            # second order
                     (pde_hip.UnitGrid([16, 16], periodic=True), {"x": "anti-periodic", "y": "periodic"}),
                     (pde_hip.UnitGrid([8, 8, 64], periodic=True), "auto_periodic_neumann")]:                        # 3-D
        eq = pde_hip.DiffusionPDE(bc=bc)
        state = pde_hip.ScalarField(grid, 0.5)
        spec = backend.make_rhs_spec(eq, state)
        a, b = DeviceArray(spec.info).set_valid(state.data), DeviceArray(spec.info)
        backend._lib.euler_multi_2d(spec.info.ref, spec.ref, a.ptr, b.ptr, 0.01, 2, C.byref(done), None)
        assert done.value == 0
        # ... and the loop still gives the right answer through the other kernels
        res = eq.solve(state, t_range=0.1, dt=0.01, solver="euler", backend="hip")
        assert np.isfinite(res.data).all()


@pytest.mark.parametrize("steps", [1, 7, 8, 9, 37, 2100])
@pytest.mark.parametrize("kind", ["diffusion", "cahn_hilliard"])
def test_euler_run_uses_it_with_remainders_and_graph_replay(kind, steps, monkeypatch):
    """pdehip_euler_run: blocks of 8 (4) steps per launch, the remainder as a shorter launch, long runs through the cached
    hipGraph — equal to the oracle's single steps; PDEHIP_TILE2D=off gives the same bits through the other kernels."""
    grid = pde_hip.UnitGrid([48, 80], periodic=[False, True])
    data = np.random.default_rng(4).uniform(-0.3, 0.3, grid.shape)
    bc = {"x": {"derivative": 0.1}, "y": "periodic"}
    if kind == "diffusion":
        eq, code, param, dt = pde_hip.DiffusionPDE(0.5, bc=bc), _abi.RHS_DIFFUSION, 0.5, 0.05
    else:
        eq, code, param, dt = pde_hip.CahnHilliardPDE(1.0, bc_c=bc, bc_mu=bc), _abi.RHS_CAHN_HILLIARD, 1.0, 1e-3
    res, info = eq.solve(pde_hip.ScalarField(grid, data), t_range=steps * dt, dt=dt, solver="euler", backend="hip", ret_info=True)
    assert info["solver"]["steps"] == steps
    np.testing.assert_array_equal(res.data, _oracle(grid, bc, bc, code, param, data, dt, steps, np.float64))
