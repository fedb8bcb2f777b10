"""GPU test of the slab-parallel path on ONE GPU (libpdehip + RCCL to self).

The gpurun box has one GPU, so the RCCL path is exercised with world size 1 and ``force_exchange=True``: the periodic
axis-0 halo then travels through ncclSend/ncclRecv to self on the halo stream, overlapped with the interior kernel on
the compute stream — the same loop templates (``csrc/pdehip_slab_loops.h``), stream/event choreography and matching
order that N > 1 ranks use.  N = 2, 3, 4 run the same templates on CPU (tests/test_distributed_gloo.py); N real GPUs:
tests/test_hip_multirank.py.  No torch in this file: the data plane is libpdehip only.
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


@pytest.mark.parametrize("force", [True, False])
@pytest.mark.parametrize("shape", [(16, 12, 128), (3, 8, 64), (2, 4, 64), (1, 4, 64), (12, 256)])
def test_diffusion_euler_overlapped_self_exchange(force, shape):
    from pde_hip.distributed import SlabStepper

    grid = pde_hip.UnitGrid(shape, periodic=True)
    data = np.random.default_rng(5).uniform(-1, 1, shape)
    eq = pde_hip.DiffusionPDE(0.8)
    st = SlabStepper(eq, grid, force_exchange=force)
    assert st.exchanging == force
    final, info = st.solve(data, t_range=1.1, dt=0.1, solver="euler")
    st.close()
    assert info["steps"] == 11
    np.testing.assert_array_equal(final, _expect(_abi.RHS_DIFFUSION, 0.8, grid, eq.bc, data, 0.1, 11))


@pytest.mark.parametrize("steps", [1, 2, 7, 12])
def test_two_steps_per_sweep_slab_loop(monkeypatch, steps):
    """pdehip_slab_euler2_run: two halo layers exchanged (to self) once per two steps, odd step counts end in a single step."""
    from pde_hip.distributed import SlabStepper

    monkeypatch.setenv("PDEHIP_SLAB_EULER4", "0")   # (16 layers would take four steps per exchange: the test below)
    grid = pde_hip.UnitGrid((16, 8, 128), periodic=[True, False, True])
    data = np.random.default_rng(6).uniform(-1, 1, grid.shape)
    eq = pde_hip.DiffusionPDE(0.7, bc={"x": "periodic", "y": {"value": 0.2}, "z": "periodic"})
    st = SlabStepper(eq, grid, force_exchange=True)
    assert st._euler2 and not st._euler4
    final, info = st.solve(data, t_range=steps * 0.05, dt=0.05, solver="euler")
    st.close()
    np.testing.assert_array_equal(final, _expect(_abi.RHS_DIFFUSION, 0.7, grid, eq.bc, data, 0.05, steps))
    # thin slabs (< 4 layers) and 2-D grids keep the one-step loop
    st1 = SlabStepper(eq, pde_hip.UnitGrid((3, 8, 64), periodic=[True, False, True]), force_exchange=True)
    assert not st1._euler2
    st1.close()
    monkeypatch.setenv("PDEHIP_SLAB_EULER2", "0")
    st2 = SlabStepper(eq, grid, force_exchange=True)
    assert not st2._euler2
    final2, _ = st2.solve(data, t_range=steps * 0.05, dt=0.05, solver="euler")
    st2.close()
    np.testing.assert_array_equal(final2, final)


@pytest.mark.parametrize("mode", ["1", "2", "3", "4"])
@pytest.mark.parametrize("shape,bc", [((16, 8, 128), {"x": "periodic", "y": {"value": 0.2}, "z": "periodic"}),
                                      ((9, 12, 256), {"x": "periodic", "y": "periodic", "z": {"derivative": -0.1}}),
                                      ((8, 6, 130), {"x": "periodic", "y": "periodic", "z": "periodic"})])
def test_four_steps_per_exchange_slab_loop(monkeypatch, mode, shape, bc):
    """pdehip_slab_euler4_run (round 6): FOUR halo layers exchanged (to self) once per four steps - the first sweep of a group computes two
    layers more per side, its boundary part waits for the exchange; every remainder of the step count; PDEHIP_SLAB_DEEP_MODE=2 also cuts
    the second sweep, 3 (the default) computes the boundary layers a group ahead on the halo stream.  Bit-identical to the serial oracle."""
    from pde_hip.distributed import SlabStepper

    monkeypatch.setenv("PDEHIP_SLAB_DEEP_MODE", mode)
    periodic = [v == "periodic" for v in bc.values()]
    grid = pde_hip.UnitGrid(shape, periodic=periodic)
    data = np.random.default_rng(6).uniform(-1, 1, grid.shape)
    eq = pde_hip.DiffusionPDE(0.7, bc=bc)
    for steps in (4, 5, 6, 7, 13, 16):
        st = SlabStepper(eq, grid, force_exchange=True)
        assert st._euler2 and st._euler4
        final, info = st.solve(data, t_range=steps * 0.05, dt=0.05, solver="euler")
        st.close()
        assert info["steps"] == steps and info["steps_per_exchange"] == 4
        np.testing.assert_array_equal(final, _expect(_abi.RHS_DIFFUSION, 0.7, grid, eq.bc, data, 0.05, steps), err_msg=f"{steps} steps")
    # fewer than 8 layers, or switched off: two steps per exchange
    st1 = SlabStepper(eq, pde_hip.UnitGrid((7,) + tuple(shape[1:]), periodic=periodic), force_exchange=True)
    assert st1._euler2 and not st1._euler4
    st1.close()
    monkeypatch.setenv("PDEHIP_SLAB_EULER4", "0")
    st2 = SlabStepper(eq, grid, force_exchange=True)
    assert st2._euler2 and not st2._euler4
    st2.close()


def test_four_steps_per_exchange_between_physical_faces():
    """The same loop on a rank WITHOUT neighbours (direct C call, lower = upper = -1): no halos, the boundary launches meet the physical
    faces of the slowest axis (index translation of the upper face for ranges that do not start at the first own layer)."""
    from pde_hip.backend import convert_bcs
    from pde_hip.device import DeviceArray, GridInfo
    from pde_hip.distributed import SlabStepper

    shape = (20, 8, 128)
    grid = pde_hip.UnitGrid(shape, periodic=False)
    pairs = [({"value": 0.4}, {"derivative": -0.2}), ({"derivative": 0.3}, {"value": -0.1}), ({"type": "mixed", "value": 0.5, "const": 0.2}, {"value": 0.0})]
    bc = {f"{a}{s}": v for a, (lo, hi) in zip(grid.axes, pairs) for s, v in (("-", lo), ("+", hi))}
    data = np.random.default_rng(9).uniform(-1, 1, shape)
    helper = SlabStepper(pde_hip.DiffusionPDE(), pde_hip.UnitGrid(shape, periodic=True), force_exchange=True)
    lib, comm = helper.lib, helper.comm
    info = GridInfo(grid.shape, grid.discretization, np.float64)
    rhs = _abi.RHS()
    rhs.kind, rhs.param = _abi.RHS_DIFFUSION, 0.6
    convert_bcs(grid.get_boundary_conditions(bc)).copy_into(rhs.bc_c)
    ok = C.c_int(0)
    lib.slab_euler4_supported(info.ref, C.byref(rhs), C.byref(ok))
    assert ok.value
    a, b = DeviceArray(info), DeviceArray(info)
    res = C.c_void_p()
    for steps in (4, 7, 10):
        a.set_valid(data)
        lib.slab_euler4_run(comm, info.ref, C.byref(rhs), -1, -1, a.ptr, b.ptr, 0.05, steps, C.byref(res), None)
        lib.stream_synchronize(None)
        np.testing.assert_array_equal((b if res.value == b.ptr else a).get_valid(), _expect(_abi.RHS_DIFFUSION, 0.6, grid, bc, data, 0.05, steps))
    helper.close()


@pytest.mark.parametrize("thick", [2, 5, 8])
def test_two_steps_per_sweep_slab_loop_with_thick_boundary_chunks(monkeypatch, thick):
    """PDEHIP_SLAB_THICK (round 5; off by default - measured slower to self, profiles/r05_probe_block.md): the first and
    the last `thick` layers as chunks of the ordinary sweep on the compute stream, the layers in between while the halo stream exchanges."""
    from pde_hip.distributed import SlabStepper

    monkeypatch.setenv("PDEHIP_SLAB_THICK", str(thick))
    monkeypatch.setenv("PDEHIP_SLAB_EULER4", "0")
    grid = pde_hip.UnitGrid((24, 8, 128), periodic=[True, False, True])
    data = np.random.default_rng(6).uniform(-1, 1, grid.shape)
    eq = pde_hip.DiffusionPDE(0.7, bc={"x": "periodic", "y": {"value": 0.2}, "z": "periodic"})
    for steps in (2, 7, 12):
        st = SlabStepper(eq, grid, force_exchange=True)
        assert st._euler2
        final, info = st.solve(data, t_range=steps * 0.05, dt=0.05, solver="euler")
        st.close()
        np.testing.assert_array_equal(final, _expect(_abi.RHS_DIFFUSION, 0.7, grid, eq.bc, data, 0.05, steps))


@pytest.mark.parametrize("solver,steps", [("euler", 5), ("runge-kutta", 3)])
@pytest.mark.parametrize("shape", [(8, 8, 128), (6, 6, 72), (12, 128)])
def test_cahn_hilliard_slab_one_sweep_per_rhs(solver, steps, shape):
    """Cahn-Hilliard on a slab: ONE exchange of two layers of c per right-hand side where the two-level kernel covers the
    grid (3-D), else two kernels with two exchanges — both bit-identical to the serial oracle."""
    from pde_hip.distributed import FUSED_CH, SlabStepper

    grid = pde_hip.UnitGrid(shape, periodic=[True] + [False] * (len(shape) - 1))
    data = np.random.default_rng(8).uniform(-0.5, 0.5, shape)
    eq = pde_hip.CahnHilliardPDE(0.9)
    st = SlabStepper(eq, grid, force_exchange=True)
    assert bool(st.flags & FUSED_CH) == (len(shape) == 3)
    final, info = st.solve(data, t_range=steps * 1e-3, dt=1e-3, solver=solver)
    st.close()
    np.testing.assert_array_equal(final, _expect(_abi.RHS_CAHN_HILLIARD, 0.9, grid, "auto_periodic_neumann", data, 1e-3, steps, solver))


@pytest.mark.parametrize("shape", [(16, 12, 128), (5, 8, 64), (12, 256)])
def test_slab_loops_on_a_rank_owning_both_physical_faces(shape):
    """The C slab loops on a rank WITHOUT neighbours (physical faces on both ends of the slowest axis): boundary layers and
    interior are separate sub-slab launches whose faces come from `sub_faces` — must equal the oracle."""
    from pde_hip.backend import convert_bcs
    from pde_hip.device import DeviceArray, GridInfo
    from pde_hip.distributed import SlabStepper

    grid = pde_hip.UnitGrid(shape, periodic=False)
    pairs = [({"value": 0.4}, {"derivative": -0.2}), ({"derivative": 0.3}, {"value": -0.1}), ({"type": "mixed", "value": 0.5, "const": 0.2}, {"value": 0.0})]
    bc = {f"{a}{s}": v for a, (lo, hi) in zip(grid.axes, pairs) for s, v in (("-", lo), ("+", hi))}
    data = np.random.default_rng(9).uniform(-1, 1, shape)
    # a communicator of size 1 (created like SlabStepper does) with lower = upper = -1
    helper = SlabStepper(pde_hip.DiffusionPDE(), pde_hip.UnitGrid(shape, periodic=True), force_exchange=True)
    lib, comm = helper.lib, helper.comm
    info = GridInfo(grid.shape, grid.discretization, np.float64)
    rhs = _abi.RHS()
    rhs.kind, rhs.param = _abi.RHS_DIFFUSION, 0.6
    faces = convert_bcs(grid.get_boundary_conditions(bc))
    faces.copy_into(rhs.bc_c)
    a, b = DeviceArray(info).set_valid(data), DeviceArray(info)
    res = C.c_void_p()
    lib.slab_euler_run(comm, info.ref, C.byref(rhs), -1, -1, a.ptr, b.ptr, 0.05, 5, C.byref(res), None)
    lib.stream_synchronize(None)
    got = (b if res.value == b.ptr else a).get_valid()
    np.testing.assert_array_equal(got, _expect(_abi.RHS_DIFFUSION, 0.6, grid, bc, data, 0.05, 5))
    ok = C.c_int(0)
    lib.slab_euler2_supported(info.ref, C.byref(rhs), C.byref(ok))
    assert bool(ok.value) == (len(shape) == 3 and shape[0] >= 4)
    if ok.value:
        a.set_valid(data)
        lib.slab_euler2_run(comm, info.ref, C.byref(rhs), -1, -1, a.ptr, b.ptr, 0.05, 5, C.byref(res), None)
        lib.stream_synchronize(None)
        np.testing.assert_array_equal((b if res.value == b.ptr else a).get_valid(), _expect(_abi.RHS_DIFFUSION, 0.6, grid, bc, data, 0.05, 5))
    helper.close()


@pytest.mark.parametrize("kind", ["diffusion", "cahn_hilliard"])
def test_rk4_and_adaptive_rkf45_in_one_c_call(kind):
    """pdehip_slab_rk4_run / pdehip_slab_rkf45_run (stage sweeps, MAX all-reduce, accept/reject and the dt controller in C):
    bit-identical final state, equal step counts and the same next dt as the serial oracle loop."""
    from pde_hip.distributed import SlabStepper

    grid = pde_hip.UnitGrid((8, 8, 128), periodic=True)
    data = np.random.default_rng(11).uniform(-0.1, 0.1, grid.shape)
    eq = pde_hip.DiffusionPDE(0.5) if kind == "diffusion" else pde_hip.CahnHilliardPDE(1.0)
    code, param = (_abi.RHS_DIFFUSION, 0.5) if kind == "diffusion" else (_abi.RHS_CAHN_HILLIARD, 1.0)
    st = SlabStepper(eq, grid, force_exchange=True)
    final, info = st.solve(data, t_range=0.01, dt=1e-3, solver="runge-kutta")
    np.testing.assert_array_equal(final, _expect(code, param, grid, "auto_periodic_neumann", data, 1e-3, 10, "runge-kutta"))
    final2, info2 = st.solve(data, t_range=0.3 if kind == "diffusion" else 0.05, dt=None, solver="runge-kutta")
    st.close()
    case = {"pde": kind, "D": 0.5, "gamma": 1.0, "bc": "auto_periodic_neumann", "t_range": 0.3 if kind == "diffusion" else 0.05, "dt": None,
            "solver": "runge-kutta"}
    expect, steps, dt_last = oracle_solve(case, grid, np.float64, data)
    assert info2["steps"] == steps and info2["attempts"] >= steps
    assert info2["dt"] == pytest.approx(dt_last, rel=1e-12)
    np.testing.assert_array_equal(final2, expect)
    stats = info2["dt_statistics"]
    assert stats["count"] == steps and 0 < stats["min"] <= stats["mean"] <= stats["max"]
    # ... and the serial product stepper (mirror API) agrees
    ref, rinfo = eq.solve(pde_hip.ScalarField(grid, data), t_range=case["t_range"], dt=None, solver="runge-kutta", backend="hip", ret_info=True)
    assert rinfo["solver"]["steps"] == steps
    assert max_rel(final2, ref.data) == 0.0


def test_adaptive_loop_reports_too_small_steps():
    """dt below dt_min -> RuntimeError with the reference's message (pde/solvers/base.py:583-590)."""
    from pde_hip.distributed import SlabStepper

    grid = pde_hip.UnitGrid((8, 8, 64), periodic=True)
    data = np.random.default_rng(12).uniform(-1, 1, grid.shape) * 1e6
    st = SlabStepper(pde_hip.CahnHilliardPDE(1.0), grid, force_exchange=True)
    with pytest.raises(RuntimeError, match="Time step below|NaN even though"):
        st.solve(data, t_range=1.0, dt=None, solver="runge-kutta", dt_min=1e-4)
    st.close()


# ---- block decomposition on ONE GPU: the faces of every periodic axis travel through pack -> RCCL to self -> unpack -----------
@pytest.mark.parametrize("shape,periodic", [((12, 8, 136), [True, True, True]), ((6, 10, 72), [True, False, True]), ((9, 7, 200), [False, True, False]),
                                            ((24, 72), [True, True]), ((5, 3, 8), [True, True, True])])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_block_layer_self_exchange(shape, periodic, dtype):
    """`BlockStepper(force_exchange=True)` at world size 1: every periodic axis exchanges its faces with the block itself (staging
    buffers, pack / unpack kernels, one RCCL group per right-hand side; rows that end inside a vector included), the stencil
    kernels read those ghost cells from memory - Euler, RK4 and the adaptive loop equal the oracle's serial run bit for bit."""
    from pde_hip.distributed import BlockStepper

    grid = pde_hip.CartesianGrid([[0, n * 0.7] for n in shape], shape, periodic=periodic)
    bc = {a: "periodic" if p else {"value": 0.3} for a, p in zip(grid.axes, periodic)}
    data = np.random.default_rng(8).uniform(-0.5, 0.5, shape).astype(dtype)
    for eq, kind, param in [(pde_hip.DiffusionPDE(0.6, bc=bc), _abi.RHS_DIFFUSION, 0.6), (pde_hip.CahnHilliardPDE(0.9, bc_c=bc, bc_mu=bc), _abi.RHS_CAHN_HILLIARD, 0.9)]:
        st = BlockStepper(eq, grid, dtype, force_exchange=True)
        assert st.exchanging and list(st.nb6)[: 2 * len(shape)] == [0 if p else -1 for p in periodic for _ in (0, 1)]
        g = oracle_grid(grid, dtype)
        hf = host_faces(grid.get_boundary_conditions(bc))
        scratch = np.zeros(grid._shape_full, dtype)
        rhs = O.make_rhs(kind, param, hf.c, hf.c, scratch)
        dt = 1e-3
        final, info = st.solve(data, t_range=5 * dt, dt=dt, solver="euler")
        np.testing.assert_array_equal(final, interior(grid, O.euler_run(g, rhs, to_full(grid, data), dt, 5)))
        final, info = st.solve(data, t_range=2 * dt, dt=dt, solver="runge-kutta")
        y = to_full(grid, data)
        for _ in range(2):
            O.rk4_step(g, rhs, y, dt)
        np.testing.assert_array_equal(final, interior(grid, y))
        if dtype == np.float64:
            case = {"bc": bc, "t_range": 0.02, "dt": None, "solver": "runge-kutta", "pde": "diffusion" if kind == _abi.RHS_DIFFUSION else "cahn_hilliard",
                    "D": param, "gamma": param}
            expect, steps, dt_last = oracle_solve(case, grid, dtype, data)
            final, info = st.solve(data, t_range=0.02, dt=None, solver="runge-kutta")
            assert info["steps"] == steps
            np.testing.assert_array_equal(final, expect)
        st.close()


@pytest.mark.parametrize("shape", [(40, 36, 256), (12, 9, 136), (7, 18, 520), (16, 16, 64), (33, 12, 128)])
@pytest.mark.parametrize("cut", [(1, 1, 0), (1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 0, 1), (1, 0, 1)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fast_block_loop_to_self(shape, cut, dtype):
    """The fast block loop (csrc/pdehip_block2_loops.h) on one device, the halos sent to the block itself through RCCL: two steps per
    sweep on the box (plain halo planes / rows, the fastest axis wrapping in the kernel), the sweep of the next pair started before the
    halos have landed, the rim recomputed behind it - bit-identical to the oracle's single steps for every combination of exchanged
    axes (a cut fastest axis: halo cells in the padding of the rows, the transposed rim kernel), rows that end inside a chunk, odd row counts (moved tiles), several x-chunks, even and odd step counts, fp64 and fp32."""
    import ctypes as C

    from pde_hip.distributed import BlockStepper

    grid = pde_hip.CartesianGrid([[0, n * 0.8] for n in shape], shape, periodic=True)
    eq = pde_hip.DiffusionPDE(0.6)
    data = np.random.default_rng(9).uniform(-0.5, 0.5, shape).astype(dtype)
    st = BlockStepper(eq, grid, dtype, force_exchange=True)
    assert st.block2 and list(st.cut) == [1, 1, 0]
    st.cut[:] = cut
    st._cut3 = (C.c_int * 3)(*st.cut)
    ok = C.c_int(0)
    st.lib.block2_supported(st.info.ref, C.byref(st._rhs2), st._cut3, C.byref(ok))
    assert ok.value == 1       # (fp32 with a cut fastest axis too since the end of round 6: the interior box on the narrow tile)
    g = oracle_grid(grid, dtype)
    hf = host_faces(grid.get_boundary_conditions("periodic"))
    rhs = O.make_rhs(_abi.RHS_DIFFUSION, 0.6, hf.c, hf.c, None)
    dt = 0.02
    for steps in (2, 6, 7):
        final, info = st.solve(data, t_range=steps * dt, dt=dt, solver="euler")
        np.testing.assert_array_equal(final, interior(grid, O.euler_run(g, rhs, to_full(grid, data), dt, steps)), err_msg=f"{steps} steps")
    st.close()


def test_block_ghost_faces_read_from_memory():
    """The kernels of a block sweep read the ghost cells of EXCHANGED faces from memory on all three axes (faces marked SKIP) and
    evaluate the physical faces on the fly: each of the 8 blocks of a 2 x 2 x 2 cut, with its ghost layers taken from the unsplit
    field, advances one Euler step to exactly its part of the unsplit result."""
    from pde_hip.device import DeviceArray, GridInfo
    from pde_hip.mesh import BlockMesh

    backend = pde_hip.get_backend("hip")
    lib = backend._lib
    grid = pde_hip.CartesianGrid([[0, 4], [0, 3], [0, 40]], [12, 10, 264], periodic=[False, True, False])
    bc = {"x-": {"value": 0.2}, "x+": {"derivative": 0.1}, "y": "periodic", "z-": {"derivative": -0.3}, "z+": {"value": 0.4}}
    bcs = grid.get_boundary_conditions(bc)
    data = np.random.default_rng(9).uniform(-0.5, 0.5, grid.shape)
    g = oracle_grid(grid)
    hf = host_faces(bcs)
    full = to_full(grid, data)
    O.set_ghost_cells(g, 1, hf.c, full)
    rhs_o = O.make_rhs(_abi.RHS_DIFFUSION, 0.7, hf.c)
    expect = interior(grid, O.euler_run(g, rhs_o, to_full(grid, data), 2e-3, 1))
    for rank in range(8):
        mesh = BlockMesh(grid, [2, 2, 2], rank)
        info = GridInfo(mesh.local_shape, grid.discretization, np.float64)
        faces = mesh.block_faces(bcs)
        rhs = _abi.RHS()
        rhs.kind, rhs.param = _abi.RHS_DIFFUSION, 0.7
        faces.copy_into(rhs.bc_c)
        window = np.ascontiguousarray(full[tuple(slice(lo, hi + 2) for lo, hi in zip(mesh.lo, mesh.hi))])
        a, b = DeviceArray(info).set_hostfull(window), DeviceArray(info)
        res = C.c_void_p()
        none6 = (C.c_int * 6)(*([-1] * 6))          # no transport here: the ghost layers are already in place
        lib.block_run(None, info.ref, C.byref(rhs), none6, 0, 0, a.ptr, b.ptr, None, None, 2e-3, 1, None, C.byref(res), None)
        got = (b if res.value == b.ptr else a).get_valid()
        np.testing.assert_array_equal(got, mesh.extract(expect), err_msg=f"block {rank} {mesh.lo}")


# ---- conditions given as expressions on slabs and blocks (refreshed by the device program inside the C loops) ------------------
_EXPR_BC = {"x": "periodic", "y-": {"value_expression": "0.2 * sin(3 * t) + 0.05 * x"}, "y+": {"derivative_expression": "0.1 * cos(t) * z - 0.3 * value**3"},
            "z-": {"virtual_point": "value / (1 + value**2) + 0.05 * x"}, "z+": {"derivative_expression": "0.05 * y * sin(t)"}}


@pytest.mark.parametrize("solver,dt", [("euler", 2e-3), ("runge-kutta", 4e-3), ("runge-kutta", None)])
def test_expression_conditions_in_the_slab_and_block_loops(solver, dt):
    """Time-dependent conditions and conditions that read the field, on a slab / block that exchanges its periodic axis with itself
    (RCCL to self): the device program runs before every right-hand side of `pdehip_slab_*_run` / `pdehip_block_run` - equal to the
    single-GPU loops (`pdehip_euler_run` ... with the same program), bit for bit, with equal step counts."""
    from pde_hip.distributed import BlockStepper, SlabStepper

    grid = pde_hip.CartesianGrid([[0, 3], [0, 2], [-1, 1]], [10, 6, 72], periodic=[True, False, False])
    data = np.random.default_rng(11).uniform(-0.5, 0.5, grid.shape)
    eq = pde_hip.DiffusionPDE(0.02, bc=_EXPR_BC)
    expect, info = eq.solve(pde_hip.ScalarField(grid, data), 0.04, dt, solver=solver, ret_info=True)
    assert np.isfinite(expect.data).all() and np.abs(expect.data - data).max() > 1e-3
    for cls in (SlabStepper, BlockStepper):
        st = cls(eq, grid, force_exchange=True)
        assert st.exchanging and st.bc_program is not None
        final, sinfo = st.solve(data, 0.04, dt, solver)
        st.close()
        assert sinfo["steps"] == info["solver"]["steps"], cls.__name__
        np.testing.assert_array_equal(final, expect.data, err_msg=cls.__name__)


def test_expression_conditions_of_a_block_use_the_coordinates_of_the_whole_grid():
    """Each of the 8 blocks of a 2 x 2 x 2 cut - faces cut to the block, `pdehip_bcprog_face_t::first` = the block's first cell - advances
    one Euler step at t = 0.37 to exactly its part of the unsplit step (conditions depending on x, y, z, t and the field)."""
    from pde_hip.bc_expr import convert_bcs_with_expressions, program_for
    from pde_hip.device import DeviceArray, GridInfo
    from pde_hip.mesh import BlockMesh

    backend = pde_hip.get_backend("hip")
    lib = backend._lib
    grid = pde_hip.CartesianGrid([[0, 4], [0, 3], [0, 40]], [12, 10, 264], periodic=False)
    bc = {"x-": {"value_expression": "0.2 * sin(3 * t) + 0.05 * y * z"}, "x+": {"derivative_expression": "0.1 * cos(t) * z - 0.3 * value**3"},
          "y-": {"virtual_point": "value / (1 + value**2) + 0.05 * x"}, "y+": {"derivative_expression": "0.05 * x * sin(t) + 0.01 * z"},
          "z-": {"value_expression": "tanh(x - y) * t"}, "z+": {"derivative": 0.4}}
    bcs = grid.get_boundary_conditions(bc)
    data = np.random.default_rng(9).uniform(-0.5, 0.5, grid.shape)
    dt, t = 2e-3, 0.37

    def one_step(info, faces, start):
        rhs = _abi.RHS()
        rhs.kind, rhs.param, rhs.t = _abi.RHS_DIFFUSION, 0.7, t
        faces.copy_into(rhs.bc_c)
        prog = program_for(lib, [faces], info)
        rhs.bc_program = prog.ptr
        a, b = start(DeviceArray(info)), DeviceArray(info)
        res = C.c_void_p()
        none6 = (C.c_int * 6)(*([-1] * 6))
        lib.block_run(None, info.ref, C.byref(rhs), none6, 0, 0, a.ptr, b.ptr, None, None, dt, 1, None, C.byref(res), None)
        return (b if res.value == b.ptr else a), a

    info_w = GridInfo(grid.shape, grid.discretization, np.float64)
    out, inp = one_step(info_w, convert_bcs_with_expressions(bcs), lambda arr: arr.set_valid(data))
    expect = out.get_valid()
    full = to_full(grid, data)      # inner faces of a block read the neighbour's cells; physical faces are evaluated by the block itself
    assert np.abs(expect - data).max() > 1e-4
    for rank in range(8):
        mesh = BlockMesh(grid, [2, 2, 2], rank)
        info = GridInfo(mesh.local_shape, grid.discretization, np.float64)
        window = np.ascontiguousarray(full[tuple(slice(lo, hi + 2) for lo, hi in zip(mesh.lo, mesh.hi))])
        got, _ = one_step(info, mesh.block_faces(bcs), lambda arr: arr.set_hostfull(window))  # noqa: B023
        np.testing.assert_array_equal(got.get_valid(), mesh.extract(expect), err_msg=f"block {rank} {mesh.lo}")


def test_fresh_buffers_are_not_overtaken_by_their_zero_fill():
    """`pdehip_malloc` zero-fills on the null stream; the steppers work on non-blocking streams.  Before the allocation waited for its
    fill, 15-24 % of such tiny runs came back all zero once the allocator recycled blocks (after ~500 iterations;
    profiles/r03_malloc_fill_race.md): many fresh steppers, upload -> 4 steps -> download, every result identical to the first."""
    from pde_hip.distributed import SlabStepper

    cases = [(pde_hip.CahnHilliardPDE(0.917), pde_hip.UnitGrid([8], periodic=True)), (pde_hip.DiffusionPDE(0.7), pde_hip.UnitGrid([8, 6], periodic=True))]
    first = {}
    for it in range(1200):
        for k, (eq, grid) in enumerate(cases):
            data = np.random.default_rng(3).uniform(-0.4, 0.4, grid.shape)
            st = SlabStepper(eq, grid)
            a, b = st.buf("state_a"), st.buf("state_b")
            st.set_local(a, data)
            assert np.array_equal(st.gather_local(a), data), (it, k)
            out = st.gather_local(st.euler_steps(a, b, 1e-3, 4))
            st.close()
            assert out.any() and np.array_equal(out, first.setdefault(k, out)), (it, k)


@pytest.mark.parametrize("dims", ["slab", "auto"])
def test_any_expression_pde_with_the_exchange_to_self(dims):
    """`DecomposedExpressionStepper(force_exchange=True)` on ONE GPU: the run-time compiled passes of expression PDEs without a fused
    decomposed loop, the ghost layers of every operator's operand through RCCL to self (slab: axis 0; blocks: every periodic axis) -
    nested operators, a two-field system, conditions of time / of the field, RK4 and the adaptive loop: equal to the single-GPU run bit
    for bit (N > 1 ranks: tests/test_distributed_gloo.py::test_any_expression_pde_on_decomposed_grids on the shim)."""
    from pde_hip.distributed import DecomposedExpressionStepper

    cases = [
        (pde_hip.PDE({"c": "laplace(c**3 - c - 0.8 * laplace(c)) + 0.1 * y"},
                     bc={"x": "periodic", "y-": {"value_expression": "0.1*sin(t) + 0.05*x"}, "y+": {"derivative_expression": "-0.2 * value**3"}, "z": "periodic"}),
         pde_hip.CartesianGrid([[0, 8], [0, 6], [0, 72]], [8, 6, 72], periodic=[True, False, True]), 0.004, 1e-3, "runge-kutta", 1),
        (pde_hip.PDE({"u": "laplace(u) + 1 - 3 * u + u**2 * v", "v": "0.1 * laplace(v + 0.2 * u) + 2 * u - u**2 * v"}, bc={"x": "periodic", "y": {"derivative": 0.1}}),
         pde_hip.UnitGrid([12, 136], periodic=[True, False]), 0.02, 0.002, "euler", 2),
        (pde_hip.PDE({"h": "0.5 * laplace(h) + 0.3 * gradient_squared(h)"}), pde_hip.UnitGrid([10, 72], periodic=True), 0.3, None, "runge-kutta", 1),
    ]
    for eq, grid, t_range, dt, solver, nfields in cases:
        data = np.random.default_rng(7).uniform(0.1, 0.6, ((nfields,) if nfields > 1 else ()) + tuple(grid.shape))
        state = pde_hip.FieldCollection([pde_hip.ScalarField(grid, d) for d in data]) if nfields > 1 else pde_hip.ScalarField(grid, data)
        expect, info = eq.solve(state, t_range, dt, solver=solver, ret_info=True)
        assert np.isfinite(expect.data).all() and np.abs(expect.data - data).max() > 1e-4
        st = DecomposedExpressionStepper(eq, state, dims=dims, force_exchange=True)
        assert st.comm is not None
        final, sinfo = st.solve(data, t_range, dt, solver)
        st.close()
        assert sinfo["steps"] == info["solver"]["steps"]
        np.testing.assert_array_equal(final, expect.data)
