"""GPU parity tests for the fused right-hand sides and explicit steppers.

`eq.solve(..., backend="hip")` through the mirror API is compared with the reference's own
solutions (tests/golden/steppers.npz) and, step by step, with the CPU oracle through the C ABI.
Tolerances: fp64 bit-exact vs oracle and vs the reference torch-CPU Euler runs; 1e-10 relative
(BASELINE.json north_star) vs the reference numpy+scipy RK4/RKF45/adaptive-Euler runs with equal
step counts; fp32 1e-5 relative vs the reference's pure-fp32 torch run.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import pytest
from helpers import case_ids, get_case, host_faces, interior, make_grid, max_rel, oracle_grid, to_full

import pde_hip
from oracle import pde_oracle as O
from pde_hip import _abi
from pde_hip.device import DeviceArray, DeviceScalar, ptr_array

pytestmark = pytest.mark.gpu

STEPS = case_ids("steppers.npz")


@pytest.fixture(scope="module")
def backend():
    return pde_hip.get_backend("hip")


def _make_eq(case):
    if case["pde"] == "diffusion":
        return pde_hip.DiffusionPDE(case["D"], bc=case["bc"])
    return pde_hip.CahnHilliardPDE(case["gamma"], bc_c=case["bc"], bc_mu=case["bc"])


@pytest.mark.parametrize("cid", STEPS)
def test_solve_vs_reference(golden_steppers, cid):
    case = get_case(golden_steppers, cid)
    grid = make_grid(case)
    dtype = np.dtype(case.get("dtype", "float64"))
    state = pde_hip.ScalarField(grid, golden_steppers[f"{cid}/input"], dtype=dtype)
    eq = _make_eq(case)
    res, info = eq.solve(state, t_range=case["t_range"], dt=case["dt"], solver=case["solver"], backend="hip", ret_info=True)
    ref = golden_steppers[f"{cid}/final"]
    assert info["solver"]["steps"] == int(golden_steppers[f"{cid}/steps"])
    np.testing.assert_allclose(info["controller"]["t_final"], float(golden_steppers[f"{cid}/t_final"]), rtol=1e-12)
    assert res.data.dtype == dtype
    if dtype == np.float32:
        assert max_rel(res.data.astype(np.float64), ref.astype(np.float64)) < 1e-5
    elif case["backend"] == "torch":
        np.testing.assert_array_equal(res.data, ref)
    else:
        assert max_rel(res.data, ref) < 1e-10
        np.testing.assert_allclose(info["solver"]["dt"], float(golden_steppers[f"{cid}/dt_last"]), rtol=1e-6)
    if case["dt"] is None:
        stats = info["solver"]["dt_statistics"]
        assert stats["count"] == info["solver"]["steps"] and stats["min"] > 0
    # the input state is not modified (Controller copies, controller.py:434)
    np.testing.assert_array_equal(state.data, golden_steppers[f"{cid}/input"].astype(dtype))


def test_expression_pde_equals_class(golden_steppers):
    """PDE({'c': 'laplace(c**3 - c - laplace(c))'}) == CahnHilliardPDE (tests/pdes/test_generic_pdes.py:27-63)."""
    cid = "ch2d_rkf45_numpy"
    case = get_case(golden_steppers, cid)
    grid = make_grid(case)
    state = pde_hip.ScalarField(grid, golden_steppers[f"{cid}/input"])
    eq = pde_hip.PDE({"c": "laplace(c**3 - c - laplace(c))"}, bc=case["bc"])
    res = eq.solve(state, t_range=case["t_range"], dt=None, solver="runge-kutta", backend="hip")
    assert max_rel(res.data, golden_steppers[f"{cid}/final"]) < 1e-10
    eq_d = pde_hip.PDE({"u": "D * laplace(u)"}, bc=case["bc"], consts={"D": 0.5})
    r1 = eq_d.evolution_rate(state).data
    r2 = pde_hip.DiffusionPDE(0.5, bc=case["bc"]).evolution_rate(state).data
    np.testing.assert_array_equal(r1, r2)
    # expressions without a hand-fused kernel go through the run-time specialised kernels (tests/test_expressions.py)
    assert np.isfinite(pde_hip.PDE({"c": "laplace(c) + c**2"}).evolution_rate(state).data).all()
    with pytest.raises(NotImplementedError, match="no kernel for operator"):
        pde_hip.PDE({"c": "curl(c)"}).evolution_rate(state)


def test_pde_rhs_vs_oracle(backend, rng):
    """make_pde_rhs == numpy-semantics evolution_rate (tests/pdes/test_generic_pdes.py:27-63)."""
    grid = pde_hip.CartesianGrid([[0, 4], [0, 6], [0, 5]], [8, 12, 10], periodic=[True, False, False])
    bc = {"x": "periodic", "y": {"value": 0.2}, "z": {"derivative": 0.1}}
    data = rng.uniform(-1, 1, grid.shape)
    state = pde_hip.ScalarField(grid, data)
    g = oracle_grid(grid)
    faces = host_faces(grid.get_boundary_conditions(bc))
    for eq, rhs in [
        (pde_hip.DiffusionPDE(0.3, bc=bc), O.make_rhs(_abi.RHS_DIFFUSION, 0.3, faces.c)),
        (pde_hip.CahnHilliardPDE(0.8, bc_c=bc, bc_mu=bc), O.make_rhs(_abi.RHS_CAHN_HILLIARD, 0.8, faces.c, faces.c, np.zeros(grid._shape_full))),
    ]:
        scratch = np.zeros(grid._shape_full)
        rhs.scratch_mu = scratch.ctypes.data
        expect = interior(grid, O.rhs_scaled(g, rhs, to_full(grid, data), 1.0))
        np.testing.assert_array_equal(eq.evolution_rate(state).data, expect)


@pytest.mark.parametrize("shape,dtype", [((32, 64), np.float64), ((8, 12, 128), np.float64), ((16, 16, 64), np.float32), ((50,), np.float64)])
@pytest.mark.parametrize("kind", ["diffusion", "cahn_hilliard"])
def test_cabi_steppers_vs_oracle(backend, rng, shape, dtype, kind):
    """pdehip_euler_run / rk4_step / rkf45_attempt through the C ABI, bit-exact against the oracle."""
    grid = pde_hip.CartesianGrid([[0, n * 0.9] for n in shape], shape, periodic=[True] + [False] * (len(shape) - 1))
    bc = "auto_periodic_neumann"
    bcs = grid.get_boundary_conditions(bc)
    data = rng.uniform(-0.5, 0.5, shape).astype(dtype)
    g = oracle_grid(grid, dtype)
    hf = host_faces(bcs)
    scratch = np.zeros(grid._shape_full, dtype)
    if kind == "diffusion":
        orhs = O.make_rhs(_abi.RHS_DIFFUSION, 0.6, hf.c)
        eq = pde_hip.DiffusionPDE(0.6, bc=bc)
    else:
        orhs = O.make_rhs(_abi.RHS_CAHN_HILLIARD, 0.9, hf.c, hf.c, scratch)
        eq = pde_hip.CahnHilliardPDE(0.9, bc_c=bc, bc_mu=bc)
    dt = 0.002
    state = pde_hip.ScalarField(grid, data, dtype=dtype)
    spec = backend.make_rhs_spec(eq, state)
    info, lib = spec.info, backend._lib

    # Euler, 7 steps; *result names the buffer holding the final state (7 single sweeps -> second buffer; grids
    # covered by the two-steps-per-sweep kernel take 3 double sweeps + 1 single -> first buffer)
    a, b = DeviceArray(info).set_valid(data), DeviceArray(info)
    res = C.c_void_p()
    lib.euler_run(info.ref, spec.ref, a.ptr, b.ptr, dt, 7, C.byref(res), None)
    assert res.value in (a.ptr, b.ptr)
    got = (b if res.value == b.ptr else a).get_valid()
    np.testing.assert_array_equal(got, interior(grid, O.euler_run(g, orhs, to_full(grid, data), dt, 7)))

    # RK4
    y = DeviceArray(info).set_valid(data)
    work = [DeviceArray(info) for _ in range(7)]
    lib.rk4_step(info.ref, spec.ref, y.ptr, ptr_array(work[:5]), dt, None)
    yo = to_full(grid, data)
    O.rk4_step(g, orhs, yo, dt)
    np.testing.assert_array_equal(y.get_valid(), interior(grid, yo))

    # RKF45 attempt
    y.set_valid(data)
    ynew, err = DeviceArray(info), DeviceScalar()
    lib.rkf45_attempt(info.ref, spec.ref, y.ptr, ynew.ptr, ptr_array(work), dt, err.ptr, None)
    yo_new, err_o = O.rkf45_attempt(g, orhs, to_full(grid, data), dt)
    np.testing.assert_array_equal(ynew.get_valid(), interior(grid, yo_new))
    assert err.value() == err_o
    np.testing.assert_array_equal(y.get_valid(), data)  # y itself is unchanged


def test_nan_propagates_to_error_norm(backend):
    """A NaN in the state must surface as error=NaN (adaptive loop shrinks dt, solvers/base.py:577-590)."""
    grid = pde_hip.UnitGrid([16, 16], periodic=True)
    data = np.zeros(grid.shape)
    data[3, 4] = np.nan
    eq = pde_hip.DiffusionPDE()
    state = pde_hip.ScalarField(grid, data)
    spec = backend.make_rhs_spec(eq, state)
    y, ynew, err = DeviceArray(spec.info).set_valid(data), DeviceArray(spec.info), DeviceScalar()
    work = [DeviceArray(spec.info) for _ in range(7)]
    backend._lib.rkf45_attempt(spec.info.ref, spec.ref, y.ptr, ynew.ptr, ptr_array(work), 1e-3, err.ptr, None)
    assert np.isnan(err.value())
    with pytest.raises(RuntimeError, match="Encountered NaN"):
        eq.solve(state, t_range=1.0, dt=None, solver="runge-kutta", backend="hip")


def test_unsupported_solver_and_pde(backend):
    grid = pde_hip.UnitGrid([8, 8])
    state = pde_hip.ScalarField(grid, 1.0)

    class ImplicitSolver(pde_hip.solvers.SolverBase):
        name = "implicit-test"

    with pytest.raises(NotImplementedError, match="does not support solver"):
        pde_hip.DiffusionPDE().solve(state, 1.0, dt=0.1, solver=ImplicitSolver(pde_hip.DiffusionPDE()))
    with pytest.raises(ValueError, match="Unknown solver"):
        pde_hip.DiffusionPDE().solve(state, 1.0, dt=0.1, solver="no-such-solver")
    # stochastic equations: Euler-Maruyama only (tests/test_noise.py); other solvers refuse, adaptive steps raise like the reference
    with pytest.raises(NotImplementedError, match="stochastic"):
        pde_hip.DiffusionPDE(noise=0.1).solve(state, 1.0, dt=0.1, solver="runge-kutta")


def test_diffusion_steady_state_and_erf(backend):
    """Known answers of the reference's solver tests.

    * heaviside initial state -> 0.5 + 0.5 erf(x/2) at t=1 within 1e-2 (tests/solvers/test_generic_solvers.py:123-148)
    * Dirichlet steady state is linear (tests/pdes/test_diffusion_pdes.py:45-70)
    """
    from scipy.special import erf

    grid = pde_hip.CartesianGrid([[-10, 10]], 100)
    x = grid.axes_coords[0]
    state = pde_hip.ScalarField(grid, (x > 0).astype(float))
    for solver, dt in [("euler", 0.005), ("runge-kutta", 0.01), ("runge-kutta", None), ("euler", None)]:
        res = pde_hip.DiffusionPDE().solve(state, t_range=1.0, dt=dt, solver=solver, backend="hip")
        np.testing.assert_allclose(res.data, 0.5 + 0.5 * erf(x / 2), atol=1e-2, rtol=1e-2)
    g2 = pde_hip.UnitGrid([16])
    res = pde_hip.DiffusionPDE(bc={"x-": {"value": 0}, "x+": {"value": 1}}).solve(pde_hip.ScalarField(g2, 0.5), t_range=400, dt=0.2, backend="hip")
    np.testing.assert_allclose(res.data, (g2.axes_coords[0]) / 16, atol=1e-3)


BC_SETS = {
    "dirichlet_all": {"value": 0.7},
    "neumann_all": {"derivative": -0.4},
    "mixed_faces": "MIXED",          # different first-order condition on every face (built per grid)
    "antiperiodic": "ANTI",
    "second_order": {"curvature": 0.3},  # not fusable: falls back to the ghost-cell kernel
    "inhomogeneous": "ARRAYS",       # per-cell values: ghost-cell kernel path
}


def _bc_for(name, grid):
    axes = grid.axes
    if BC_SETS[name] == "MIXED":
        conds = [{"value": 0.5}, {"derivative": 0.25}, {"type": "mixed", "value": 1.5, "const": 0.2}, {"value": -0.3}, {"derivative": -1.0}, {"type": "mixed", "value": -0.5, "const": 1.0}]
        return {f"{a}{s}": conds[2 * i + k] for i, a in enumerate(axes) for k, s in enumerate("-+")}
    if BC_SETS[name] == "ANTI":
        return {a: "anti-periodic" for a in axes}
    if BC_SETS[name] == "ARRAYS":
        rng = np.random.default_rng(5)
        bc = {}
        for i, a in enumerate(axes):
            face = tuple(n for j, n in enumerate(grid.shape) if j != i)
            bc[a + "-"] = {"value": rng.uniform(-1, 1, face)}
            bc[a + "+"] = {"derivative": rng.uniform(-1, 1, face)}
        return bc
    return BC_SETS[name]


@pytest.mark.parametrize("shape", [(6, 10, 128), (5, 7, 200), (4, 6, 520), (3, 3, 1032), (9, 256), (5, 600), (2, 2, 4), (1, 1, 8), (12, 16, 64)])
@pytest.mark.parametrize("bc_name", list(BC_SETS))
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_on_the_fly_bcs_vs_oracle(backend, shape, bc_name, dtype):
    """Every BC family x tile geometry (row end inside / at / beyond a wave tile, 1-cell axes):
    the stencil kernel's on-the-fly BCs == ghost-cell kernel + stencil of the oracle, bit-exact,
    for both right-hand sides (Euler steps chain 1-2 fused kernels per step)."""
    if dtype == np.float32 and shape[-1] % 4:
        pytest.skip("fp32 vector path needs a multiple of 4 cells")
    periodic = bc_name == "antiperiodic"
    if bc_name == "second_order" and min(shape) < 2:
        pytest.skip("curvature BC needs 2 support points")
    grid = pde_hip.CartesianGrid([[0, n * 0.8] for n in shape], shape, periodic=periodic)
    bc = _bc_for(bc_name, grid)
    bcs = grid.get_boundary_conditions(bc)
    data = np.random.default_rng(11).uniform(-0.5, 0.5, shape).astype(dtype)
    g = oracle_grid(grid, dtype)
    hf = host_faces(bcs)
    state = pde_hip.ScalarField(grid, data, dtype=dtype)
    for eq, orhs in [
        (pde_hip.DiffusionPDE(0.6, bc=bc), O.make_rhs(_abi.RHS_DIFFUSION, 0.6, hf.c)),
        (pde_hip.CahnHilliardPDE(0.9, bc_c=bc, bc_mu=bc), O.make_rhs(_abi.RHS_CAHN_HILLIARD, 0.9, hf.c, hf.c, np.zeros(grid._shape_full, dtype))),
    ]:
        scratch = np.zeros(grid._shape_full, dtype)
        orhs.scratch_mu = scratch.ctypes.data
        spec = backend.make_rhs_spec(eq, state)
        a, b = DeviceArray(spec.info).set_valid(data), DeviceArray(spec.info)
        res = C.c_void_p()
        backend._lib.euler_run(spec.info.ref, spec.ref, a.ptr, b.ptr, 1e-3, 3, C.byref(res), None)
        got = (b if res.value == b.ptr else a).get_valid()
        np.testing.assert_array_equal(got, interior(grid, O.euler_run(g, orhs, to_full(grid, data), 1e-3, 3)))
        k = DeviceArray(spec.info)
        a.set_valid(data)
        backend._lib.rhs_scaled(spec.info.ref, spec.ref, a.ptr, k.ptr, 0.01, None)
        np.testing.assert_array_equal(k.get_valid(), interior(grid, O.rhs_scaled(g, orhs, to_full(grid, data), 0.01)))


@pytest.mark.parametrize("shape", [(6, 10, 128), (5, 8, 128), (4, 8, 200), (5, 7, 200), (4, 6, 520), (9, 256), (5, 600), (2, 2, 4), (1, 1, 8), (7, 5, 3), (33,)])
@pytest.mark.parametrize("bc_name", ["periodic", "mixed_faces", "second_order", "inhomogeneous"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind", ["diffusion", "cahn_hilliard"])
def test_runge_kutta_stage_sweeps_vs_oracle(backend, shape, bc_name, dtype, kind):
    """RK4 steps and RKF45 attempts whose stages run as ONE sweep each (slope + the pointwise combination that follows,
    error norm included) - diffusion on the vectorised one-level kernel, Cahn-Hilliard on the two-level kernel (fp64, and
    fp32 in 2-D) - and as separate kernels elsewhere (odd rows, 1-D, faces the two-level kernel does not cover): state,
    new state and error norm equal the oracle's lincomb / combine sequence bit for bit, for on-the-fly faces, ghost-cell
    faces (curvature, per-cell values) and row ends inside / at / beyond a wave tile."""
    if bc_name == "second_order" and min(shape) < 2:
        pytest.skip("curvature BC needs 2 support points")
    grid = pde_hip.CartesianGrid([[0, n * 0.8] for n in shape], shape, periodic=bc_name == "periodic")
    bc = "periodic" if bc_name == "periodic" else _bc_for(bc_name, grid)
    data = np.random.default_rng(13).uniform(-0.5, 0.5, shape).astype(dtype)
    g = oracle_grid(grid, dtype)
    hf = host_faces(grid.get_boundary_conditions(bc))
    scratch = np.zeros(grid._shape_full, dtype)
    if kind == "diffusion":
        orhs = O.make_rhs(_abi.RHS_DIFFUSION, 0.6, hf.c)
        eq = pde_hip.DiffusionPDE(0.6, bc=bc)
    else:
        orhs = O.make_rhs(_abi.RHS_CAHN_HILLIARD, 0.9, hf.c, hf.c, scratch)
        eq = pde_hip.CahnHilliardPDE(0.9, bc_c=bc, bc_mu=bc)
    spec = backend.make_rhs_spec(eq, pde_hip.ScalarField(grid, data, dtype=dtype))
    info, lib = spec.info, backend._lib
    dt = 2e-3
    y = DeviceArray(info).set_valid(data)
    work = [DeviceArray(info) for _ in range(7)]
    yo = to_full(grid, data)
    for _ in range(3):
        lib.rk4_step(info.ref, spec.ref, y.ptr, ptr_array(work[:5]), dt, None)
        O.rk4_step(g, orhs, yo, dt)
    np.testing.assert_array_equal(y.get_valid(), interior(grid, yo))
    ynew, err = DeviceArray(info), DeviceScalar()
    for _ in range(2):   # second attempt starts from the first one's result (work arrays are dirty)
        start = y.get_valid()
        lib.rkf45_attempt(info.ref, spec.ref, y.ptr, ynew.ptr, ptr_array(work), dt, err.ptr, None)
        yo_new, err_o = O.rkf45_attempt(g, orhs, to_full(grid, start), dt)
        np.testing.assert_array_equal(ynew.get_valid(), interior(grid, yo_new))
        assert err.value() == err_o
        np.testing.assert_array_equal(y.get_valid(), start)
        y, ynew = ynew, y


@pytest.mark.parametrize("kind,shape", [("diffusion", (12, 16)), ("diffusion", (6, 8, 128)), ("cahn_hilliard", (8, 8, 64)), ("cahn_hilliard", (24,))])
def test_adams_bashforth_vs_oracle(backend, rng, kind, shape):
    """Two-step Adams-Bashforth (rate of the previous step kept instead of re-evaluated) == oracle, bit-exact; a second
    call of the stepper continues the multi-step history like the reference's closure does."""
    grid = pde_hip.UnitGrid(shape, periodic=[True] + [False] * (len(shape) - 1))
    bc = "auto_periodic_neumann"
    data = rng.uniform(-0.3, 0.3, shape)
    hf = host_faces(grid.get_boundary_conditions(bc))
    g = oracle_grid(grid)
    scratch = np.zeros(grid._shape_full)
    if kind == "diffusion":
        eq, orhs = pde_hip.DiffusionPDE(0.7, bc=bc), O.make_rhs(_abi.RHS_DIFFUSION, 0.7, hf.c)
    else:
        eq, orhs = pde_hip.CahnHilliardPDE(0.8, bc_c=bc, bc_mu=bc), O.make_rhs(_abi.RHS_CAHN_HILLIARD, 0.8, hf.c, hf.c, scratch)
    dt = 2e-3
    res, info = eq.solve(pde_hip.ScalarField(grid, data), t_range=9 * dt, dt=dt, solver="adams-bashforth", backend="hip", ret_info=True)
    assert info["solver"]["steps"] == 9
    np.testing.assert_array_equal(res.data, interior(grid, O.adams_bashforth_run(g, orhs, to_full(grid, data), dt, 9)))
    # two calls of one stepper (tracker interrupts) == one call over the whole range
    solver = pde_hip.solvers.AdamsBashforthSolver(eq)
    state = pde_hip.ScalarField(grid, data)
    stepper = solver.make_stepper(state, dt)
    t = stepper(state, 0.0, 4 * dt)
    stepper(state, t, 9 * dt)
    np.testing.assert_array_equal(state.data, res.data)


@pytest.mark.parametrize("kind,shape", [("diffusion", (16, 128)), ("diffusion", (8, 8, 64)), ("cahn_hilliard", (12, 72)), ("diffusion", (10, 7))])
def test_euler_run_through_the_captured_graph(backend, rng, kind, shape):
    """Long runs on small grids replay a captured hipGraph of 32 steps (16 double sweeps where the two-level kernel covers
    the grid): 2100 steps = 65 replays + 20 plain steps, then a second call re-uses the cached graph - bit-exact."""
    grid = pde_hip.UnitGrid(shape, periodic=[True] + [False] * (len(shape) - 1))
    bc = "auto_periodic_neumann"
    data = rng.uniform(-0.3, 0.3, shape)
    hf = host_faces(grid.get_boundary_conditions(bc))
    g = oracle_grid(grid)
    scratch = np.zeros(grid._shape_full)
    if kind == "diffusion":
        eq, orhs, dt = pde_hip.DiffusionPDE(0.7, bc=bc), O.make_rhs(_abi.RHS_DIFFUSION, 0.7, hf.c), 0.05
    else:
        eq, orhs, dt = pde_hip.CahnHilliardPDE(0.8, bc_c=bc, bc_mu=bc), O.make_rhs(_abi.RHS_CAHN_HILLIARD, 0.8, hf.c, hf.c, scratch), 1e-3
    spec = backend.make_rhs_spec(eq, pde_hip.ScalarField(grid, data))
    a, b = DeviceArray(spec.info).set_valid(data), DeviceArray(spec.info)
    res = C.c_void_p()
    lib = backend._lib
    lib.euler_run(spec.info.ref, spec.ref, a.ptr, b.ptr, dt, 2100, C.byref(res), None)
    got = (b if res.value == b.ptr else a).get_valid()
    full = O.euler_run(g, orhs, to_full(grid, data), dt, 2100)
    np.testing.assert_array_equal(got, interior(grid, full))
    # the same buffers, grid, rhs and dt again: the cached graph (64 more steps = 2 replays); the state continues from `a`
    a.set_valid(got)
    lib.euler_run(spec.info.ref, spec.ref, a.ptr, b.ptr, dt, 64, C.byref(res), None)
    np.testing.assert_array_equal((b if res.value == b.ptr else a).get_valid(), interior(grid, O.euler_run(g, orhs, full, dt, 64)))


@pytest.mark.parametrize("kind,shape", [("diffusion", (16, 128)), ("cahn_hilliard", (8, 8, 64)), ("diffusion", (10, 7))])
def test_rk4_run_through_the_captured_graph(backend, rng, kind, shape):
    """pdehip_rk4_run: 530 RK4 steps on a small grid = 66 replays of a captured 8-step graph + 2 plain steps, then the
    cached graph again - equal to 530 (+ 17) single steps of the oracle, bit for bit."""
    grid = pde_hip.UnitGrid(shape, periodic=[True] + [False] * (len(shape) - 1))
    bc = "auto_periodic_neumann"
    data = rng.uniform(-0.3, 0.3, shape)
    hf = host_faces(grid.get_boundary_conditions(bc))
    g = oracle_grid(grid)
    scratch = np.zeros(grid._shape_full)
    if kind == "diffusion":
        eq, orhs, dt = pde_hip.DiffusionPDE(0.7, bc=bc), O.make_rhs(_abi.RHS_DIFFUSION, 0.7, hf.c), 0.05
    else:
        eq, orhs, dt = pde_hip.CahnHilliardPDE(0.8, bc_c=bc, bc_mu=bc), O.make_rhs(_abi.RHS_CAHN_HILLIARD, 0.8, hf.c, hf.c, scratch), 1e-3
    spec = backend.make_rhs_spec(eq, pde_hip.ScalarField(grid, data))
    y = DeviceArray(spec.info).set_valid(data)
    work = [DeviceArray(spec.info) for _ in range(5)]
    lib = backend._lib
    lib.rk4_run(spec.info.ref, spec.ref, y.ptr, ptr_array(work), dt, 530, None)
    yo = to_full(grid, data)
    for _ in range(530):
        O.rk4_step(g, orhs, yo, dt)
    np.testing.assert_array_equal(y.get_valid(), interior(grid, yo))
    lib.rk4_run(spec.info.ref, spec.ref, y.ptr, ptr_array(work), dt, 17, None)   # cached graph: 2 replays + 1 step
    for _ in range(17):
        O.rk4_step(g, orhs, yo, dt)
    np.testing.assert_array_equal(y.get_valid(), interior(grid, yo))
    # and through the solver front end (fixed-step Runge-Kutta = one rk4_run per stepper call)
    res = eq.solve(pde_hip.ScalarField(grid, data), t_range=20 * dt, dt=dt, solver="runge-kutta", adaptive=False, backend="hip", tracker=None)
    yo = to_full(grid, data)
    for _ in range(20):
        O.rk4_step(g, orhs, yo, dt)
    np.testing.assert_array_equal(res.data, interior(grid, yo))


def test_full_size_runge_kutta_sweeps_512cubed(backend):
    """BASELINE size (512^3 fp64, periodic diffusion): one RK4 step and one RKF45 attempt through the one-sweep-per-stage
    path.  Slabs of the results equal the oracle run on those slabs alone with 6 spare layers per side (every stage moves
    an error at the slab end one layer inwards: 4 resp. 6 layers) - bit for bit; the periodic sum is conserved; the error
    estimate is finite and positive."""
    n, dt = 512, 0.05
    grid = pde_hip.UnitGrid([n, n, n], periodic=True)
    u = np.random.default_rng(1).random((n, n, n))
    eq = pde_hip.DiffusionPDE(1.0)
    spec = backend.make_rhs_spec(eq, pde_hip.ScalarField(grid, u))
    info, lib = spec.info, backend._lib
    y, ynew, err = DeviceArray(info).set_valid(u), DeviceArray(info), DeviceScalar()
    work = [DeviceArray(info) for _ in range(7)]
    lib.rkf45_attempt(info.ref, spec.ref, y.ptr, ynew.ptr, ptr_array(work), dt, err.ptr, None)
    got45, e45 = ynew.get_valid(), err.value()
    lib.rk4_step(info.ref, spec.ref, y.ptr, ptr_array(work[:5]), dt, None)
    got4 = y.get_valid()
    pad, keep = 6, 4
    for lo in (0, 253, n - keep):
        idx = np.arange(lo - pad, lo + keep + pad) % n
        sub = pde_hip.CartesianGrid([[0, len(idx)], [0, n], [0, n]], [len(idx), n, n], periodic=[False, True, True])
        bcs = sub.get_boundary_conditions({"x": {"derivative": 0}, "y": "periodic", "z": "periodic"})
        rhs = O.make_rhs(_abi.RHS_DIFFUSION, 1.0, host_faces(bcs).c)
        g = oracle_grid(sub)
        new45, _ = O.rkf45_attempt(g, rhs, to_full(sub, u[idx]), dt)
        np.testing.assert_array_equal(got45[lo:lo + keep], interior(sub, new45)[pad:pad + keep])
        full4 = to_full(sub, u[idx])
        O.rk4_step(g, rhs, full4, dt)
        np.testing.assert_array_equal(got4[lo:lo + keep], interior(sub, full4)[pad:pad + keep])
    assert np.isfinite(e45) and e45 > 0
    assert abs(got45.sum() - u.sum()) < 1e-9 * u.sum() and abs(got4.sum() - u.sum()) < 1e-9 * u.sum()
