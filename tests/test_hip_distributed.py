"""GPU test of the slab stepper's product engine (HipEngine: libpdehip + torch streams + RCCL).

Only one GPU is available to the test box, so the RCCL path is exercised with world size 1 and
`force_exchange=True`: the periodic axis-0 halo then travels through ncclSend/ncclRecv to self on
the halo stream, overlapped with the interior kernel on the compute stream — the same code path,
stream/event choreography and P2P ordering that N > 1 ranks use (N = 2, 3 are covered bit-exactly
on CPU with gloo in test_distributed_gloo.py).
"""

from __future__ import annotations

import os
import socket

import numpy as np
import pytest
from helpers import host_faces, interior, oracle_grid, to_full

import pde_hip
from oracle import pde_oracle as O
from pde_hip import _abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def process_group():
    import torch
    import torch.distributed as dist

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def _expect(eq_kind, param, grid, bc, data, dt, steps, solver="euler"):
    g = oracle_grid(grid)
    hf = host_faces(grid.get_boundary_conditions(bc))
    scratch = np.zeros(grid._shape_full)
    rhs = O.make_rhs(eq_kind, param, hf.c, hf.c, scratch)
    y = to_full(grid, data)
    if solver == "euler":
        y = O.euler_run(g, rhs, y, dt, steps)
    else:
        for _ in range(steps):
            O.rk4_step(g, rhs, y, dt)
    return interior(grid, y)


@pytest.mark.parametrize("comm_mode", ["native", "torch"])
@pytest.mark.parametrize("force", [True, False])
@pytest.mark.parametrize("shape", [(16, 12, 128), (3, 8, 64), (2, 4, 64), (1, 4, 64), (12, 256)])
def test_diffusion_euler_overlapped_self_exchange(process_group, monkeypatch, comm_mode, force, shape):
    """native = libpdehip's own RCCL communicator + C step loop; torch = torch.distributed P2P ops."""
    from pde_hip.distributed import HipEngine, SlabStepper

    monkeypatch.setenv("PDEHIP_COMM", comm_mode)

    grid = pde_hip.UnitGrid(shape, periodic=[True] + [False] * (len(shape) - 1))
    data = np.random.default_rng(2).uniform(-1, 1, shape)
    eq = pde_hip.DiffusionPDE(0.8)
    st = SlabStepper(eq, grid, engine=HipEngine(0), force_exchange=force)
    assert st.exchanging == force
    assert (st.comm is not None) == (force and comm_mode == "native")
    final, info = st.solve(data, t_range=1.1, dt=0.1, solver="euler")
    assert info["steps"] == 11
    np.testing.assert_array_equal(final, _expect(_abi.RHS_DIFFUSION, 0.8, grid, eq.bc, data, 0.1, 11))


@pytest.mark.parametrize("shape,periodic,steps", [
    ((16, 12, 128), [True, False, False], 11),   # 5 double sweeps + 1 single step, local faces on y and z
    ((4, 8, 256), [True, True, True], 6),        # interior sweep empty (4 layers = 2 + 2 boundary layers)
    ((9, 6, 128), [True, True, False], 2),       # 2-row tiles, one double sweep
    ((40, 8, 128), [True, False, True], 7),
])
def test_diffusion_euler_two_steps_per_sweep_slab_loop(process_group, monkeypatch, shape, periodic, steps):
    """The slab loop with two halo layers (exchange every other step) == serial oracle, bit-exact."""
    from pde_hip.distributed import HipEngine, SlabStepper

    monkeypatch.setenv("PDEHIP_COMM", "native")
    grid = pde_hip.UnitGrid(shape, periodic=periodic)
    data = np.random.default_rng(4).uniform(-1, 1, shape)
    eq = pde_hip.DiffusionPDE(0.7)
    st = SlabStepper(eq, grid, engine=HipEngine(0), force_exchange=True)
    assert st.comm is not None and st._euler2
    final, info = st.solve(data, t_range=steps * 0.05, dt=0.05, solver="euler")
    assert info["steps"] == steps
    np.testing.assert_array_equal(final, _expect(_abi.RHS_DIFFUSION, 0.7, grid, eq.bc, data, 0.05, steps))
    # grids the kernel does not cover keep the one-step loop
    st1 = SlabStepper(eq, pde_hip.UnitGrid((8, 7, 64), periodic=True), engine=HipEngine(0), force_exchange=True)
    assert not st1._euler2


@pytest.mark.parametrize("shape,periodic", [((8, 8, 64), [True, True, True]), ((6, 12, 128), [True, False, False]), ((2, 4, 72), [True, False, True])])
def test_cahn_hilliard_slab_one_sweep_per_rhs(process_group, monkeypatch, shape, periodic):
    """Slab Cahn-Hilliard with the fused sweep: ONE exchange (two layers of c) per right-hand side, mu never exchanged;
    Euler and RK4 equal the serial oracle bit for bit."""
    from pde_hip.distributed import HipEngine, SlabStepper

    monkeypatch.setenv("PDEHIP_COMM", "native")
    grid = pde_hip.UnitGrid(shape, periodic=periodic)
    data = np.random.default_rng(6).uniform(-0.2, 0.2, shape)
    eq = pde_hip.CahnHilliardPDE(0.9)
    for solver, steps in [("euler", 5), ("runge-kutta", 3)]:
        st = SlabStepper(eq, grid, engine=HipEngine(0), force_exchange=True)
        assert st._ch_rhs is not None
        final, info = st.solve(data, t_range=steps * 1e-3, dt=1e-3, solver=solver)
        assert info["steps"] == steps
        np.testing.assert_array_equal(final, _expect(_abi.RHS_CAHN_HILLIARD, 0.9, grid, eq.bc_c, data, 1e-3, steps, solver))


@pytest.mark.parametrize("shape", [(7, 6, 128), (2, 4, 64), (5, 256)])
def test_slab_loop_with_physical_faces_on_both_ends(process_group, shape):
    """The one-step C slab loop on a rank that owns BOTH physical faces of the slowest axis (no neighbours): boundary
    layers and interior are separate sub-slab launches whose faces come from `sub_faces` - must equal the oracle."""
    import ctypes as C

    from pde_hip.backend import convert_bcs
    from pde_hip.device import DeviceArray, GridInfo
    from pde_hip.distributed import HipEngine

    eng = HipEngine(0)
    comm = eng.make_comm(process_group, None, 1, 0)
    grid = pde_hip.UnitGrid(shape, periodic=False)
    bc = {f"{a}{s}": v for a, (lo, hi) in zip(grid.axes, [({"value": 0.4}, {"derivative": -0.2}), ({"derivative": 0.3}, {"value": -0.1}),
                                                           ({"type": "mixed", "value": 0.5, "const": 0.2}, {"value": 0.0})]) for s, v in (("-", lo), ("+", hi))}
    data = np.random.default_rng(9).uniform(-1, 1, shape)
    info = GridInfo(grid.shape, grid.discretization, np.float64)
    rhs = _abi.RHS()
    rhs.kind, rhs.param = _abi.RHS_DIFFUSION, 0.6
    faces = convert_bcs(grid.get_boundary_conditions(bc))
    faces.copy_into(rhs.bc_c)
    a, b = DeviceArray(info).set_valid(data), DeviceArray(info)
    res = C.c_void_p()
    eng.lib.slab_euler_run(comm, info.ref, C.byref(rhs), -1, -1, a.ptr, b.ptr, 0.05, 5, C.byref(res), None)
    eng.lib.stream_synchronize(None)
    got = (b if res.value == b.ptr else a).get_valid()
    np.testing.assert_array_equal(got, _expect(_abi.RHS_DIFFUSION, 0.6, grid, bc, data, 0.05, 5))
    # the two-steps-per-sweep loop on the same rank: two-ended boundary sweep with the physical faces + interior sweep
    ok = C.c_int(0)
    eng.lib.slab_euler2_supported(info.ref, C.byref(rhs), C.byref(ok))
    assert bool(ok.value) == (len(shape) == 3 and shape[0] >= 4)
    if ok.value:
        a.set_valid(data)
        eng.lib.slab_euler2_run(comm, info.ref, C.byref(rhs), -1, -1, a.ptr, b.ptr, 0.05, 5, C.byref(res), None)
        eng.lib.stream_synchronize(None)
        np.testing.assert_array_equal((b if res.value == b.ptr else a).get_valid(), _expect(_abi.RHS_DIFFUSION, 0.6, grid, bc, data, 0.05, 5))
    # Cahn-Hilliard sweep on arrays with two halo layers per side, physical faces on both ends (no exchange)
    if len(shape) == 3:
        ch = _abi.RHS()
        ch.kind, ch.param = _abi.RHS_CAHN_HILLIARD, 0.9
        faces.copy_into(ch.bc_c)
        faces.copy_into(ch.bc_mu)
        eng.lib.slab_ch_supported(info.ref, C.byref(ch), C.byref(ok))
        assert bool(ok.value) == (shape[0] >= 4)   # a rank that owns both faces needs 4 layers (never the case with > 1 rank)
    if len(shape) == 3 and ok.value:
        ext = GridInfo((shape[0] + 2, *shape[1:]), grid.discretization, np.float64)
        padded = np.zeros((shape[0] + 2, *shape[1:]))
        padded[1:-1] = data
        ce, oe = DeviceArray(ext).set_valid(padded), DeviceArray(ext)
        eng.lib.slab_ch_sweep(comm, info.ref, C.byref(ch), -1, -1, ce.ptr, oe.ptr, 1e-3, 1, None)
        eng.lib.stream_synchronize(None)
        np.testing.assert_array_equal(oe.get_valid()[1:-1], _expect(_abi.RHS_CAHN_HILLIARD, 0.9, grid, bc, data, 1e-3, 1))
    eng.lib.comm_destroy(comm)


@pytest.mark.parametrize("comm_mode", ["native", "torch"])
def test_cahn_hilliard_rk_and_adaptive_self_exchange(process_group, monkeypatch, comm_mode):
    from pde_hip.distributed import HipEngine, SlabStepper

    monkeypatch.setenv("PDEHIP_COMM", comm_mode)

    grid = pde_hip.UnitGrid([8, 8, 64], periodic=True)
    data = np.random.default_rng(3).uniform(-0.1, 0.1, grid.shape)
    eq = pde_hip.CahnHilliardPDE(1.0)
    st = SlabStepper(eq, grid, engine=HipEngine(0), force_exchange=True)
    final, info = st.solve(data, t_range=0.01, dt=1e-3, solver="runge-kutta")
    np.testing.assert_array_equal(final, _expect(_abi.RHS_CAHN_HILLIARD, 1.0, grid, eq.bc_c, data, 1e-3, 10, "runge-kutta"))
    # adaptive RKF45 == the single-GPU backend's adaptive solve (same controller, same kernels)
    st2 = SlabStepper(eq, grid, engine=HipEngine(0), force_exchange=True)
    final2, info2 = st2.solve(data, t_range=0.05, dt=None, solver="runge-kutta")
    ref, rinfo = eq.solve(pde_hip.ScalarField(grid, data), t_range=0.05, dt=None, solver="runge-kutta", backend="hip", ret_info=True)
    assert info2["steps"] == rinfo["solver"]["steps"]
    np.testing.assert_array_equal(final2, ref.data)
