"""GPU evidence for the SURVEY §8 f-rows whose logic lives on the host side of the C ABI (VERDICT r2 "weak #2"):

* f2 — expression / time-dependent boundary conditions: ghost cells at several times against the formula of the reference
  (``pde/grids/boundaries/local.py:849-866``) evaluated directly with numpy, and Euler / RK4 / adaptive RKF45 runs against a
  host loop that refreshes the faces before every right-hand side and evaluates it with the oracle's ghost setter + stencil;
* f3 — post-step hooks (``pde/backends/numba/_solvers.py:22-64``): in-place hooks, hooks that return data, ``StopIteration``
  after an in-place change;
* f4 — the state stays resident on the device across tracker interrupts (upload / download counts).

Everything runs through the mirror front end (``pde_hip.*``: the GPU box has no py-pde) and the real ``libpdehip.so``.
"""

from __future__ import annotations

import numpy as np
import pytest
from helpers import HostBuf, interior, oracle_grid, to_full

import pde_hip
from oracle import pde_oracle as O
from pde_hip.bc_expr import convert_bcs_with_expressions
from pde_hip.solvers import make_dt_adjuster

pytestmark = pytest.mark.gpu


def _wall(grid, axis, upper):
    """Wall-point coordinates of a face as arrays of the face's shape, one per axis (independent of the product code)."""
    coords = []
    for a in range(grid.num_axes):
        lo, hi = grid.axes_bounds[a]
        n = grid.shape[a]
        coords.append(np.array([hi if upper else lo]) if a == axis else lo + (np.arange(n) + 0.5) * (hi - lo) / n)
    mesh = np.meshgrid(*coords, indexing="ij")
    return [np.take(m, 0, axis=axis) for m in mesh]


# the reference's virtual-point formulas per target (pde/grids/boundaries/local.py:849-866) as numpy functions of
# (e = value of the expression, c = value of `const`, v = adjacent field value, dx)
_VIRTUAL = {
    "virtual_point": lambda e, c, v, dx: e,
    "value_expression": lambda e, c, v, dx: 2 * e - v,
    "derivative_expression": lambda e, c, v, dx: dx * e + v,
    "mixed_expression": lambda e, c, v, dx: (2 * dx * c + (2 - e * dx) * v) / (e * dx + 2),
}

FACES_2D = {
    "x-": ("value_expression", "0.3 * sin(2 * t) + 0.1 * y", None),
    "x+": ("derivative_expression", "0.2 * cos(t) * y", None),
    "y-": ("mixed_expression", "0.5 + 0.1 * x + 0.2 * t", "0.3 * sin(t + x)"),
    "y+": ("virtual_point", "0.9 * value + 0.05 * t * x", None),
}


def _bc_from(faces):
    bc = {}
    for key, (kind, expr, const) in faces.items():
        bc[key] = {"type": kind, "value": expr, "const": const} if const is not None else {kind: expr}
    return bc


def _numpy_expr(grid, text):
    import sympy

    names = ["value", "dx", *grid.axes, "t"]
    return sympy.lambdify([sympy.Symbol(n) for n in names], sympy.sympify(text, locals={n: sympy.Symbol(n) for n in names}), modules="numpy")


def _expected_ghosts(grid, faces, full, t):
    """Ghost layers of `full` (compact host layout) from the reference's formulas, face by face."""
    out = full.copy()
    nd = grid.num_axes
    for key, (kind, expr, const) in faces.items():
        axis, upper = grid.axes.index(key[0]), key[1] == "+"
        dx = float(grid.discretization[axis])
        n = grid.shape[axis]
        sl = [slice(1, -1)] * nd
        sl[axis] = n if upper else 1
        v = full[tuple(sl)]
        coords = _wall(grid, axis, upper)
        e = np.broadcast_to(_numpy_expr(grid, expr)(v, dx, *coords, t), v.shape)
        c = np.broadcast_to(_numpy_expr(grid, const)(v, dx, *coords, t), v.shape) if const is not None else 0.0
        sl[axis] = n + 1 if upper else 0
        out[tuple(sl)] = _VIRTUAL[kind](e, c, v, dx)
    return out


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_expression_ghost_cells_at_several_times(rng, dtype):
    grid = pde_hip.CartesianGrid([[0, 2], [-1, 2]], [12, 10])
    bc = _bc_from(FACES_2D)
    data = rng.uniform(-1, 1, grid.shape).astype(dtype)
    field = pde_hip.ScalarField(grid, data, dtype=dtype)
    for t in (0.0, 0.37, 1.9, 12.5):
        field.set_ghost_cells(bc, args={"t": t})
        got = field._data_full
        expect = _expected_ghosts(grid, FACES_2D, to_full(grid, data.astype(np.float64)), t)
        mask = np.ones(got.shape, bool)
        mask[0, 0] = mask[0, -1] = mask[-1, 0] = mask[-1, -1] = False      # corners are not defined
        tol = 1e-13 if dtype == np.float64 else 2e-6
        np.testing.assert_allclose(got[mask], expect[mask], rtol=tol, atol=tol)
    with pytest.raises(RuntimeError, match="Require value for `t`"):
        field.set_ghost_cells(bc)      # the reference's contract (local.py:1139-1146)
    # operators take `args` as well
    lap = field.laplace(bc, args={"t": 0.37}).data
    g = oracle_grid(grid, dtype)
    full = _expected_ghosts(grid, FACES_2D, to_full(grid, data.astype(np.float64)), 0.37).astype(dtype)
    np.testing.assert_allclose(lap, O.laplace(g, full), rtol=1e-12 if dtype == np.float64 else 1e-4, atol=1e-12 if dtype == np.float64 else 1e-4)


FACES_NONLINEAR = {
    # conditions that are NOT affine in the adjacent value (VERDICT r2 "next" #9): the coefficient arrays are rewritten from the
    # field the conditions are applied to (pde_hip/bc_expr.py: `reads_value`)
    "x-": ("derivative_expression", "-0.3 * value**3 + 0.1 * y", None),              # radiation-type flux
    "x+": ("value_expression", "0.2 * tanh(value) + 0.1 * sin(t)", None),
    "y-": ("virtual_point", "value / (1 + value**2) + 0.05 * x", None),
    "y+": ("mixed_expression", "0.5 + 0.2 * value**2", "0.1 * cos(t)"),
}


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_nonlinear_conditions_read_the_field(rng, dtype):
    """Ghost cells and an operator with conditions that depend non-linearly on the adjacent value: against the reference's
    formulas evaluated with numpy on the same field (two different fields through ONE setter: the arrays are refreshed)."""
    grid = pde_hip.CartesianGrid([[0, 2], [-1, 2]], [12, 10])
    bc = _bc_from(FACES_NONLINEAR)
    tol = 1e-13 if dtype == np.float64 else 2e-6
    mask = np.ones(tuple(n + 2 for n in grid.shape), bool)
    mask[0, 0] = mask[0, -1] = mask[-1, 0] = mask[-1, -1] = False
    for t in (0.0, 0.8):
        data = rng.uniform(-1, 1, grid.shape).astype(dtype)
        field = pde_hip.ScalarField(grid, data, dtype=dtype)
        field.set_ghost_cells(bc, args={"t": t})
        expect = _expected_ghosts(grid, FACES_NONLINEAR, to_full(grid, data.astype(np.float64)), t)
        np.testing.assert_allclose(field._data_full[mask], expect[mask], rtol=tol, atol=tol)
        lap = field.laplace(bc, args={"t": t}).data
        np.testing.assert_allclose(lap, O.laplace(oracle_grid(grid, dtype), expect.astype(dtype)), rtol=1e-12 if dtype == np.float64 else 1e-4,
                                   atol=1e-12 if dtype == np.float64 else 1e-4)
    # refused where the field the conditions would read is never stored: conditions of mu, intermediate fields of nested operators
    state = pde_hip.ScalarField(grid, rng.uniform(-1, 1, grid.shape))
    # conditions of mu that depend non-linearly on mu: the fused class right-hand side (mu in registers) declines, the expression form
    # of the class takes over - against the two operators composed on the host
    host_c, host_mu = _HostRhs(grid, {"x": {"derivative": 0.1}, "y": {"value": 0.2}}, 1.0), _HostRhs(grid, bc, 1.0)
    y0 = rng.uniform(-0.5, 0.5, grid.shape)
    y, dt_ = y0.copy(), 2e-6      # (explicit steps of a fourth-order operator: dt << dx**4)
    for i in range(5):
        y = y + dt_ * host_mu(y**3 - y - 0.8 * host_c(y, i * dt_), i * dt_)
    out = pde_hip.CahnHilliardPDE(0.8, bc_c={"x": {"derivative": 0.1}, "y": {"value": 0.2}}, bc_mu=bc).solve(
        pde_hip.ScalarField(grid, y0), t_range=5 * dt_, dt=dt_, solver="euler", backend="hip", tracker=None)
    assert np.abs(out.data - y).max() <= 1e-11 * np.abs(y).max()
    # an operator of a nested expression applies the conditions to an INTERMEDIATE field (`value` = the adjacent value of c**3 - c):
    # refreshed pass by pass from the pass's own input - against the same right-hand side composed on the host
    host = _HostRhs(grid, bc, 0.3)
    y0 = rng.uniform(-0.5, 0.5, grid.shape)
    dt_, y = 2e-4, None
    y = y0.copy()
    for i in range(6):
        y = y + dt_ * (host(y**3 - y, i * dt_) - 0.1 * y)       # 0.3 * laplace(c**3 - c) - 0.1 * c
    for solver, tol in (("euler", 1e-12), ("runge-kutta", 1e-3)):
        out = pde_hip.PDE({"c": "0.3 * laplace(c**3 - c) - 0.1 * c"}, bc=bc).solve(pde_hip.ScalarField(grid, y0), t_range=6 * dt_, dt=dt_, solver=solver,
                                                                              backend="hip", tracker=None)
        assert np.abs(out.data - y).max() <= tol * np.abs(y).max()


class _HostRhs:
    """D * laplace(y) with the expression faces refreshed at t: the product's coefficient arrays on the HOST + oracle kernels."""

    def __init__(self, grid, bc, D, dtype=np.float64, cubic=0.0):
        self.grid, self.D, self.cubic = grid, D, cubic
        self.g = oracle_grid(grid, dtype)
        self.table = convert_bcs_with_expressions(grid.get_boundary_conditions(bc), upload=HostBuf)

    def __call__(self, y, t):
        full = to_full(self.grid, y)
        self.table.update({"t": t}, state=full)   # (conditions that are not affine in the value read it from `full`)
        O.set_ghost_cells(self.g, 1, self.table.c, full)
        return self.D * O.laplace(self.g, full) - self.cubic * y * y * y


FACES_3D = {
    "x-": ("value_expression", "0.2 * sin(3 * t) + 0.05 * y", None),
    "x+": ("derivative_expression", "0.1 * cos(t) * z", None),
    "z-": ("value_expression", "0.1 * t", None),
    "z+": ("derivative_expression", "0.05 * x * sin(t)", None),
}


@pytest.mark.parametrize("nonlinear,expression", [(False, False), (True, False), (True, True)], ids=["affine", "nonlinear", "nonlinear-expression-pde"])
@pytest.mark.parametrize("shape", [(16, 24), (8, 6, 128)])
@pytest.mark.parametrize("scheme", ["euler", "rk4", "rkf45"])
def test_time_dependent_bcs_in_the_steppers(rng, shape, scheme, nonlinear, expression):
    """DiffusionPDE with time-dependent faces: every right-hand side sees the faces of ITS time (Euler t_n; RK4 t, t + dt/2,
    t + dt; RKF45 the six stage times) - and, for conditions that are not affine in the adjacent value, of ITS input field (the
    Runge-Kutta stage inputs) - against the host loop, equal step counts."""
    if len(shape) == 2:
        grid, faces = pde_hip.CartesianGrid([[0, 2], [0, 3]], shape), {k: v for k, v in FACES_2D.items() if v[0] != "virtual_point"}
        faces["y+"] = ("value_expression", "0.1 * x * cos(2 * t)", None)
        if nonlinear:
            faces = dict(FACES_NONLINEAR)
    else:
        grid, faces = pde_hip.CartesianGrid([[0, 1], [0, 1], [0, 8]], shape, periodic=[False, True, False]), dict(FACES_3D)
        if nonlinear:
            faces["x-"] = ("derivative_expression", "-0.4 * value**3 + 0.05 * y * sin(3 * t)", None)
            faces["z+"] = ("value_expression", "0.1 * tanh(2 * value) + 0.02 * x", None)
    bc = _bc_from(faces)
    if len(shape) == 3:
        bc["y"] = "periodic"
    D = 0.02 if len(shape) == 3 else 0.4
    y0 = rng.uniform(-0.5, 0.5, shape)
    state = pde_hip.ScalarField(grid, y0)
    if expression:   # a generic expression PDE (run-time built kernels, pdehip_jit_euler_run / pdehip_jit_rk_run) instead of the class
        eq = pde_hip.PDE({"c": f"{D} * laplace(c) - 0.3 * c**3"}, bc=bc)
        rhs = _HostRhs(grid, bc, D, cubic=0.3)
    else:
        eq = pde_hip.DiffusionPDE(D, bc=bc)
        rhs = _HostRhs(grid, bc, D)
    dt, nsteps = 2e-3, 9
    if scheme == "euler":
        res, info = eq.solve(state, t_range=nsteps * dt, dt=dt, solver="euler", backend="hip", ret_info=True)
        y = y0.copy()
        for i in range(nsteps):
            y = y + dt * rhs(y, i * dt)
        steps = nsteps
    elif scheme == "rk4":
        res, info = eq.solve(state, t_range=nsteps * dt, dt=dt, solver="runge-kutta", backend="hip", ret_info=True)
        y = y0.copy()
        for i in range(nsteps):
            t = i * dt
            k1 = dt * rhs(y, t)
            k2 = dt * rhs(y + 0.5 * k1, t + 0.5 * dt)
            k3 = dt * rhs(y + 0.5 * k2, t + 0.5 * dt)
            k4 = dt * rhs(y + k3, t + dt)
            y = y + (k1 + 2 * k2 + 2 * k3 + k4) / 6
        steps = nsteps
    else:
        t_end = 0.05
        res, info = eq.solve(state, t_range=t_end, dt=1e-3, solver="runge-kutta", adaptive=True, tolerance=1e-5, backend="hip", ret_info=True)
        A = [0.0, 1 / 4, 3 / 8, 12 / 13, 1.0, 1 / 2]
        B = [[1 / 4], [3 / 32, 9 / 32], [1932 / 2197, -7200 / 2197, 7296 / 2197], [439 / 216, -8.0, 3680 / 513, -845 / 4104],
             [-8 / 27, 2.0, -3544 / 2565, 1859 / 4104, -11 / 40]]
        adjust = make_dt_adjuster(1e-10, 1e10)
        y, t, dt_opt, steps = y0.copy(), 0.0, 1e-3, 0
        while True:
            h = max(min(dt_opt, t_end - t), 1e-10)
            ks = [h * rhs(y, t)]
            for s, b in enumerate(B):
                arg = y.copy()
                for bj, kj in zip(b, ks):
                    arg = arg + bj * kj
                ks.append(h * rhs(arg, t + A[s + 1] * h))
            err = np.abs(ks[0] / 360 - 128 / 4275 * ks[2] - 2197 / 75240 * ks[3] + ks[4] / 50 + 2 / 55 * ks[5]).max() / 1e-5
            if err <= 1:
                y = y + 25 / 216 * ks[0] + 1408 / 2565 * ks[2] + 2197 / 4104 * ks[3] - ks[4] / 5
                t += h
                steps += 1
            if t < t_end:
                dt_opt = adjust(h, err)
            else:
                break
    assert info["solver"]["steps"] == steps
    assert np.abs(res.data - y).max() <= 1e-12 * max(1.0, np.abs(y).max())
    assert np.abs(res.data - y0).max() > 1e-3          # the run did something


TWO_STEP_CASES = {
    # name: (bounds, shape, periodic, bc) - conditions of time and position on some faces, constants / periodic axes on others
    "2d-all-faces": ([[0, 2], [0, 3]], (37, 70), False, {"x-": {"value_expression": "0.3 * sin(2 * t) + 0.1 * y"}, "x+": {"derivative_expression": "0.2 * cos(t) * y"},
                                                        "y-": {"type": "mixed_expression", "value": "0.5 + 0.1 * x + 0.2 * t", "const": "0.3 * sin(t + x)"},
                                                        "y+": {"value_expression": "0.1 * x * cos(2 * t)"}}),
    "2d-mixed-with-constants": ([[0, 1], [0, 1]], (64, 128), False, {"x-": {"value": 0.2}, "x+": {"derivative_expression": "0.1 * t * y"}, "y": {"derivative": -0.1}}),
    "2d-periodic-axis": ([[0, 1], [0, 2]], (20, 256), [True, False], {"x": "periodic", "y-": {"value_expression": "sin(3 * t) * x"}, "y+": {"value": 0.3}}),
    "3d-all-faces": ([[0, 1]] * 3, (12, 10, 130), False, {"x-": {"value_expression": "0.2 * sin(3 * t) + 0.05 * y"}, "x+": {"derivative_expression": "0.1 * cos(t) * z"},
                                                       "y-": {"value_expression": "x * z * (1 + t)"}, "y+": {"derivative_expression": "0.05 * x * sin(t)"},
                                                       "z-": {"value_expression": "tanh(x - y) * t"}, "z+": {"derivative_expression": "0.1 * y * cos(2 * t)"}}),
    "3d-periodic-rows": ([[0, 1], [0, 1], [0, 8]], (9, 8, 128), [False, True, False], {"x-": {"value_expression": "0.2 * sin(3 * t) + 0.05 * y"}, "x+": {"derivative": 0.1},
                                                                                    "y": "periodic", "z-": {"value_expression": "0.1 * t"}, "z+": {"derivative_expression": "0.05 * x * sin(t)"}}),
    # (a periodic axis shorter than the tile of the recomputing kernel: its boxes reach beyond the images they may read)
    "3d-short-periodic-rows": ([[0, 1], [0, 1], [0, 8]], (8, 6, 128), [False, True, False], {"x-": {"value_expression": "0.2 * sin(3 * t) + 0.05 * y"}, "x+": {"derivative_expression": "0.1 * cos(t) * z"},
                                                                                          "y": "periodic", "z-": {"value_expression": "0.1 * t"}, "z+": {"derivative_expression": "0.05 * x * sin(t)"}}),
    # rows that end at chunk boundaries (the streaming sweep without the ragged-row code), several chunks per row and tiles per column
    "3d-aligned-all-faces": ([[0, 1]] * 3, (12, 8, 256), False, {"x-": {"value_expression": "0.2 * sin(3 * t) + 0.05 * y"}, "x+": {"derivative_expression": "0.1 * cos(t) * z"},
                                                              "y-": {"value_expression": "x * z * (1 + t)"}, "y+": {"derivative_expression": "0.05 * x * sin(t)"},
                                                              "z-": {"value_expression": "tanh(x - y) * t"}, "z+": {"derivative_expression": "0.1 * y * cos(2 * t)"}}),
    "3d-aligned-interior-tiles": ([[0, 1], [0, 2], [0, 3]], (6, 16, 384), False, {"x-": {"derivative_expression": "0.1 * cos(t) * z"}, "x+": {"value_expression": "0.2 * sin(3 * t) + 0.05 * y"},
                                                                                 "y-": {"derivative_expression": "0.05 * x * sin(t) + 0.1 * z"}, "y+": {"value_expression": "x * z * (1 + t)"},
                                                                                 "z-": {"derivative_expression": "0.1 * y * cos(2 * t)"}, "z+": {"value_expression": "tanh(x - y) * t"}}),
    "3d-aligned-two-chunks": ([[0, 1], [0, 2], [0, 4]], (10, 12, 256), False, {"x": {"value": 0.1}, "y-": {"type": "mixed_expression", "value": "0.5 + 0.1 * x + 0.2 * t", "const": "0.3 * sin(t + z)"},
                                                                            "y+": {"derivative": -0.2}, "z-": {"value": 0.3}, "z+": {"value_expression": "0.1 * x * cos(2 * t) + y"}}),
    "3d-aligned-periodic-march": ([[0, 1], [0, 1], [0, 4]], (16, 8, 256), [True, False, False], {"x": "periodic", "y-": {"derivative_expression": "0.3 * sin(x + 2 * t) * z"},
                                                                                             "y+": {"value_expression": "0.2 * z * t"}, "z-": {"derivative_expression": "0.1 * x * y"},
                                                                                             "z+": {"value_expression": "cos(3 * t + y)"}}),
    # 8 M cells and more, one row / eight columns beyond whole tiles / chunks: the sweep over a whole number of tiles, the rows and columns behind
    # them recomputed with the stand-in faces (shell_open_rows), THEN the two layers next to every face given as arrays with the true coefficients
    "3d-open-rows-and-columns": ([[0, 1], [0, 2], [0, 2]], (40, 513, 520), False, {"x-": {"value_expression": "0.2 * sin(3 * t) + 0.05 * y"}, "x+": {"derivative": 0.1},
                                                                              "y-": {"derivative_expression": "0.05 * x * sin(t)"}, "y+": {"value_expression": "x * z * (1 + t)"},
                                                                              "z-": {"value": 0.3}, "z+": {"derivative_expression": "0.1 * y * cos(2 * t)"}}),
    "3d-position-only": ([[0, 1]] * 3, (8, 12, 64), False, {"x": {"value": 0.1}, "y-": {"value_expression": "sin(3 * x) * z"}, "y+": {"derivative": 0.0}, "z": {"derivative_expression": "0.2 * x - y"}}),
}


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("case", sorted(TWO_STEP_CASES))
def test_two_steps_per_sweep_with_conditions_of_time_and_position(rng, case, dtype, monkeypatch):
    """Conditions that depend on time and position ride the two-step sweep (second coefficient set for t + dt, the two layers of cells
    next to such faces recomputed: pdehip_shell.hip): bit-identical to one step per sweep with a refresh and a ghost pass per step
    (PDEHIP_TIMED_TWO_STEP=0), odd and even step counts; and the same against the host loop over the oracle's operators."""
    bounds, shape, periodic, bc = TWO_STEP_CASES[case]
    grid = pde_hip.CartesianGrid(bounds, shape, periodic=periodic)
    D = 0.3 * float(min(grid.discretization)) ** 2 / 2e-3      # dt * D / dx^2 = 0.3
    y0 = rng.uniform(-0.5, 0.5, shape).astype(dtype)
    dt = 2e-3
    for nsteps in (8, 7):
        runs = []
        for env in ("1", "0"):
            monkeypatch.setenv("PDEHIP_TIMED_TWO_STEP", env)
            eq = pde_hip.DiffusionPDE(D, bc=bc)
            res = eq.solve(pde_hip.ScalarField(grid, y0, dtype=dtype), t_range=nsteps * dt, dt=dt, solver="euler", backend="hip", tracker=None)
            runs.append(res.data.copy())
        assert np.array_equal(runs[0], runs[1]), f"{case} {nsteps} steps: max diff {np.abs(runs[0] - runs[1]).max()}"
        assert np.isfinite(runs[0]).all() and np.abs(runs[0] - y0).max() > 1e-4
    if dtype == np.float64:
        rhs = _HostRhs(grid, bc, D)
        y = y0.astype(np.float64)
        for i in range(7):
            y = y + dt * rhs(y, i * dt)
        assert np.abs(runs[0] - y).max() <= 1e-12 * max(1.0, np.abs(y).max())


def test_two_steps_per_sweep_with_position_dependent_faces_in_a_replayed_graph(rng, monkeypatch):
    """Faces given as arrays that nothing rewrites (conditions of the position only), a run long enough for the captured block of steps
    (pdehip_euler_run replays a hipGraph on small grids): the two-step sweep + the cells next to the faces inside the graph."""
    grid = pde_hip.CartesianGrid([[0, 1], [0, 2]], (48, 96))
    bc = {"x-": {"value_expression": "sin(3 * y)"}, "x+": {"derivative": 0.1}, "y-": {"derivative_expression": "0.2 * x"}, "y+": {"value_expression": "x * (1 - x)"}}
    y0 = rng.uniform(-0.5, 0.5, grid.shape)
    D, dt, nsteps = 0.02, 2e-3, 203
    runs = []
    for env in ("1", "0"):
        monkeypatch.setenv("PDEHIP_TIMED_TWO_STEP", env)
        res = pde_hip.DiffusionPDE(D, bc=bc).solve(pde_hip.ScalarField(grid, y0), t_range=nsteps * dt, dt=dt, solver="euler", backend="hip", tracker=None)
        runs.append(res.data.copy())
    assert np.array_equal(runs[0], runs[1]), np.abs(runs[0] - runs[1]).max()
    rhs = _HostRhs(grid, bc, D)
    y = y0.copy()
    for i in range(nsteps):
        y = y + dt * rhs(y, 0.0)
    assert np.abs(runs[0] - y).max() <= 1e-12


# ----------------------------------------------------------------------------------------------------------------------
# f3: post-step hooks
# ----------------------------------------------------------------------------------------------------------------------
class _ClippedDiffusion(pde_hip.DiffusionPDE):
    """Diffusion whose hook clips the field in place and counts the clipped cells (the pattern of the reference's
    tests/pdes/test_pde_class.py:546-566 / tests/test_integration.py hooks)."""

    def __init__(self, *args, stop_at=None, returns=True, **kwargs):
        super().__init__(*args, **kwargs)
        self.stop_at, self.returns = stop_at, returns

    def make_post_step_hook(self, state):
        stop_at, returns = self.stop_at, self.returns

        def hook(state_data, t, post_step_data):
            mask = state_data > 0.6
            state_data[mask] = 0.6
            post_step_data = post_step_data + int(mask.sum())
            if stop_at is not None and t >= stop_at:
                state_data[0] = -1.0           # changes made before StopIteration are the final state
                raise StopIteration
            return (state_data, post_step_data) if returns else None

        return hook, 0


def _oracle_euler_step(grid, D, bc, y, dt):
    import pde_hip._abi as _abi
    from helpers import host_faces

    g = oracle_grid(grid)
    faces = host_faces(grid.get_boundary_conditions(bc))
    rhs = O.make_rhs(_abi.RHS_DIFFUSION, D, faces.c)
    return interior(grid, O.euler_run(g, rhs, to_full(grid, y), dt, 1)).copy()


@pytest.mark.parametrize("shape", [(24, 32), (6, 8, 64)])
def test_post_step_hook_runs_after_every_step(rng, shape):
    grid = pde_hip.UnitGrid(shape, periodic=[True] + [False] * (len(shape) - 1))
    bc = "auto_periodic_neumann"
    y0 = rng.uniform(0, 1, shape)
    eq = _ClippedDiffusion(0.8, bc=bc)
    res, info = eq.solve(pde_hip.ScalarField(grid, y0), t_range=1.0, dt=0.1, solver="euler", backend="hip", ret_info=True)
    y, clipped = y0.copy(), 0
    for _ in range(10):
        y = _oracle_euler_step(grid, 0.8, bc, y, 0.1)
        clipped += int((y > 0.6).sum())
        y[y > 0.6] = 0.6
    np.testing.assert_array_equal(res.data, y)
    assert info["solver"]["steps"] == 10 and info["solver"]["post_step_data"] == clipped > 0


def test_post_step_hook_stop_iteration_keeps_in_place_changes(rng):
    grid = pde_hip.UnitGrid([16, 16])
    y0 = rng.uniform(0, 1, grid.shape)
    eq = _ClippedDiffusion(0.5, stop_at=0.25)
    state = pde_hip.ScalarField(grid, y0)
    stepper = pde_hip.solvers.EulerSolver(eq, backend="hip").make_stepper(state, 0.1)
    with pytest.raises(StopIteration):
        stepper(state, 0.0, 1.0)
    y = y0.copy()
    for _ in range(4):                      # hook times 0.0, 0.1, 0.2, 0.3 (the time the step started at): stops in the 4th
        y = _oracle_euler_step(grid, 0.5, "auto_periodic_neumann", y, 0.1)
        y[y > 0.6] = 0.6
    y[0] = -1.0
    np.testing.assert_array_equal(state.data, y)


def test_post_step_hook_with_adaptive_runge_kutta(rng):
    """The hook of an adaptive run sees every ACCEPTED step with the new time (pde/backends/numba/_solvers.py:262-270)."""
    grid = pde_hip.UnitGrid([32, 16], periodic=True)
    y0 = rng.uniform(0, 1, grid.shape)
    times = []

    class Eq(pde_hip.DiffusionPDE):
        def make_post_step_hook(self, state):
            def hook(state_data, t, data):
                times.append(t)
                state_data *= 0.999
                return state_data, data + 1

            return hook, 0

    res, info = Eq(1.0).solve(pde_hip.ScalarField(grid, y0), t_range=0.5, dt=1e-3, solver="runge-kutta", adaptive=True, backend="hip", ret_info=True)
    assert info["solver"]["post_step_data"] == info["solver"]["steps"] == len(times) > 3
    assert times == sorted(times) and abs(times[-1] - 0.5) < 1e-12
    assert np.isfinite(res.data).all() and res.data.mean() < y0.mean()      # the damping of the hook


# ----------------------------------------------------------------------------------------------------------------------
# f4: residency across tracker interrupts
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("solver,adaptive", [("euler", False), ("runge-kutta", False), ("runge-kutta", True)])
def test_state_resident_across_tracker_interrupts(rng, solver, adaptive):
    grid = pde_hip.UnitGrid([48, 40, 64], periodic=[True, False, True])
    y0 = rng.uniform(0, 1, grid.shape)
    eq = pde_hip.CahnHilliardPDE(1.0)
    kw = dict(dt=1e-3, solver=solver, backend="hip", ret_info=True)
    if adaptive:
        kw["adaptive"] = True
    ref, iref = eq.solve(pde_hip.ScalarField(grid, y0), t_range=0.02, **kw)

    # (a) a tracker that never reads the data: ONE upload, ONE download (the caller's read at the end), 5 interrupts
    calls = []
    from pde_hip.solvers import Controller, SolverBase

    def run(tracker):
        s_kw = {"adaptive": True} if adaptive else {}
        sol = SolverBase.from_name(solver, pde=eq, backend="hip", **s_kw)
        ctrl = Controller(sol, t_range=0.02, tracker=tracker, interval=0.004)
        final = ctrl.run(pde_hip.ScalarField(grid, y0), 1e-3)
        return final, ctrl

    final, ctrl = run(lambda s, t: calls.append(t))
    link = final.__dict__["_hip_link"]
    assert len(calls) >= 6
    assert (link.uploads, link.downloads) == (1, 0)
    data = np.array(final.data)                       # first read: the one download
    assert (link.uploads, link.downloads) == (1, 1)
    if adaptive:
        assert np.abs(data - ref.data).max() < 1e-3   # interrupts cut steps short: another step sequence within the tolerance (1e-4 per step)
    else:
        np.testing.assert_array_equal(data, ref.data)
        assert ctrl.diagnostics["solver"]["steps"] == iref["solver"]["steps"]

    # (b) a tracker that reads AND modifies the state at every interrupt: one download and one upload per interrupt
    seen = []

    def tracker(s, t):
        seen.append(float(s.data.mean()))
        s.data[0, 0, 0] += 1e-3

    final2, _ = run(tracker)
    link2 = final2.__dict__["_hip_link"]
    n = len(seen)
    # the first interrupt comes before the first upload; the last one after the last stepper call
    assert n >= 6 and link2.uploads == n - 1 and link2.downloads == n - 1
    assert abs(seen[-1] - seen[0]) < 1e-3             # Cahn-Hilliard conserves the mean (up to the tracker's nudges)
    assert not np.array_equal(np.array(final2.data), data)


# ----------------------------------------------------------------------------------------------------------------------
# Runge-Kutta loops of expression PDEs as ONE C call (pdehip_jit_rk_run), conditions refreshed on the device
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("adaptive", [False, True], ids=["rk4", "rkf45"])
@pytest.mark.parametrize("shape", [(24, 16), (8, 6, 128)])
def test_expression_runge_kutta_loops_in_one_c_call(rng, shape, adaptive, monkeypatch):
    """RK4 / adaptive RKF45 of generic expression right-hand sides: the C loop (stage epilogues where the kernels carry them,
    time-dependent faces refreshed by the device program before every stage) equals the Python-driven loop bit for bit."""
    import pde_hip.expr as expr_mod

    nd = len(shape)
    periodic = [True] + [False] * (nd - 1)
    grid = pde_hip.CartesianGrid([[0, n * 0.5] for n in shape], shape, periodic=periodic)
    last = grid.axes[-1]
    bc = {a: "periodic" if p else {"derivative": 0} for a, p in zip(grid.axes, periodic)}
    bc_t = dict(bc)
    bc_t[last + "-"], bc_t[last + "+"] = {"value_expression": "0.2 * cos(3 * t) + 0.01 * x"}, {"derivative": 0}
    bc_t.pop(last)
    u0 = rng.uniform(-0.4, 0.4, shape)
    state = pde_hip.ScalarField(grid, u0)
    cases = {
        "allen_cahn": (pde_hip.PDE({"c": "laplace(c) - c**3 + c"}, bc=bc), state),
        "chain": (pde_hip.PDE({"c": "0.2 * c - laplace(laplace(c)) - c**3 - 2 * laplace(c)"}, bc=bc), state),
        "time_dependent": (pde_hip.PDE({"c": "laplace(c) + 0.1 * sin(t)"}, bc=bc_t), state),
        "system": (pde_hip.PDE({"u": "0.5 * laplace(u) + 1 - 4 * u + u**2 * v", "v": "0.1 * laplace(v) + 3 * u - u**2 * v"}, bc=bc),
                   pde_hip.FieldCollection([pde_hip.ScalarField(grid, 1.0 + 0.1 * u0), pde_hip.ScalarField(grid, 3.0 - 0.1 * u0)])),
    }
    calls = []
    real = expr_mod._run_rk
    monkeypatch.setattr(expr_mod, "_run_rk", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    for name, (eq, st) in cases.items():
        kw = dict(t_range=0.01, dt=5e-4, solver="runge-kutta", backend="hip", ret_info=True)
        if adaptive:
            kw["adaptive"] = True
        n0 = len(calls)
        res_c, info_c = eq.solve(st, **kw)
        assert len(calls) == n0 + 1, name
        monkeypatch.setenv("PDEHIP_EXPR_LOOP", "0")
        res_py, info_py = eq.solve(st, **kw)
        monkeypatch.delenv("PDEHIP_EXPR_LOOP")
        assert info_c["solver"]["steps"] == info_py["solver"]["steps"] > 0, name
        np.testing.assert_array_equal(res_c.data, res_py.data, err_msg=name)
        assert not np.array_equal(res_c.data, st.data)


def test_class_pde_loops_stay_in_c_with_time_dependent_bcs(rng):
    """DiffusionPDE / CahnHilliardPDE with conditions that depend on time: the right-hand-side descriptor carries a device
    program and the C loops (Euler incl. hipGraph-free path, RK4 run, the adaptive loop) follow it - against single C-ABI
    evaluations with the faces refreshed from the HOST tables (the round-2 path)."""
    import ctypes as C

    from pde_hip.device import DeviceArray, ptr_array

    grid = pde_hip.CartesianGrid([[0, 1], [0, 1], [0, 8]], (8, 6, 128), periodic=[False, True, False])
    bc = _bc_from(FACES_3D)
    bc["y"] = "periodic"
    backend = pde_hip.get_backend("hip")
    lib = backend._lib
    y0 = rng.uniform(-0.5, 0.5, grid.shape)
    state = pde_hip.ScalarField(grid, y0)
    for eq in (pde_hip.DiffusionPDE(0.02, bc=bc), pde_hip.CahnHilliardPDE(0.5, bc_c=bc, bc_mu=bc)):
        spec = backend.make_rhs_spec(eq, state)
        assert spec.program is not None and spec.c.bc_program
        info = spec.info
        a, b = DeviceArray(info).set_valid(y0), DeviceArray(info)
        res = C.c_void_p()
        dt, n = 1e-4, 7
        spec.c.t = 0.3
        lib.euler_run(info.ref, spec.ref, a.ptr, b.ptr, dt, n, C.byref(res), None)
        got = (b if res.value == b.ptr else a).get_valid()
        # the same steps one by one, every step told its time explicitly
        a.set_valid(y0)
        cur, nxt = a, b
        for i in range(n):
            spec.c.t = 0.3 + i * dt
            lib.euler_run(info.ref, spec.ref, cur.ptr, nxt.ptr, dt, 1, C.byref(res), None)
            cur, nxt = nxt, cur
        np.testing.assert_array_equal(got, cur.get_valid())
        # ... and the faces really moved: the same run at another start time differs
        a.set_valid(y0)
        spec.c.t = 1.7
        lib.euler_run(info.ref, spec.ref, a.ptr, b.ptr, dt, n, C.byref(res), None)
        assert not np.array_equal(got, (b if res.value == b.ptr else a).get_valid())
        # RK4 run == RK4 steps with explicit times
        work = [DeviceArray(info) for _ in range(5)]
        y = DeviceArray(info).set_valid(y0)
        spec.c.t = 0.3
        lib.rk4_run(info.ref, spec.ref, y.ptr, ptr_array(work), dt, 3, None)
        y2 = DeviceArray(info).set_valid(y0)
        for i in range(3):
            spec.c.t = 0.3 + i * dt
            lib.rk4_step(info.ref, spec.ref, y2.ptr, ptr_array(work), dt, None)
        np.testing.assert_array_equal(y.get_valid(), y2.get_valid())


def test_finite_check_on_the_device(rng):
    """`backend.make_finite_check()` (pdehip_count_nonfinite): the ConsistencyTracker's check as a device reduction - fields with
    NaN / +-inf anywhere (first / last cell, inside ragged rows, in one component only) are caught, finite ones pass; a resident
    state is checked without a download."""
    import ctypes as C

    from pde_hip.device import DeviceArray, DeviceBuffer

    backend = pde_hip.get_backend("hip")
    is_finite = backend.make_finite_check()
    for shape, dtype in [((7, 5, 203), np.float64), ((33, 40), np.float32), ((129,), np.float64), ((64, 64, 64), np.float32)]:
        grid = pde_hip.UnitGrid(shape)
        info = backend.grid_info(grid, dtype)
        data = rng.uniform(-1, 1, shape).astype(dtype)
        arr = DeviceArray(info).set_valid(data)
        assert is_finite(arr)
        for where, bad in [((0,) * len(shape), np.nan), (tuple(n - 1 for n in shape), np.inf), (tuple(n // 2 for n in shape), -np.inf)]:
            d2 = data.copy()
            d2[where] = bad
            assert not is_finite(arr.set_valid(d2))
        # counts per component
        two = DeviceArray(info, (2,)).set_valid(np.stack([data, np.where(data > 0.9, np.nan, data)]))
        out = DeviceBuffer(16)
        backend._lib.count_nonfinite(info.ref, 2, two.ptr, out.ptr, None)
        host = np.empty(2)
        backend._lib.memcpy_d2h(host.ctypes.data, out.ptr, 16, None)
        assert host[0] == 0 and host[1] == int((data > 0.9).sum())
    # a resident state: checked where it lives
    grid = pde_hip.UnitGrid([32, 32], periodic=True)
    seen = []
    res = pde_hip.DiffusionPDE().solve(pde_hip.ScalarField(grid, rng.uniform(0, 1, grid.shape)), t_range=1.0, dt=0.1,
                                       tracker=lambda s, t: seen.append(is_finite(s)), interval=0.2, backend="hip")
    link = res.__dict__["_hip_link"]
    assert all(seen) and len(seen) >= 5 and link.downloads == 0


def test_constants_reach_the_runtime_compiler_as_literals(rng):
    """`sin(2*pi*t)` in a condition or in the equation used to be printed as `sin(2*M_PI*t)`; hiprtc sources include no
    <math.h>.  Written with `pi` and with the literal 3.141592653589793 the runs are the same, bit for bit."""
    grid = pde_hip.UnitGrid([10, 12, 64], periodic=[False, True, True])
    y0 = rng.uniform(0, 1, grid.shape)
    runs = []
    for const in ("pi", "3.141592653589793"):
        eq = pde_hip.DiffusionPDE(0.4, bc={"x-": {"value_expression": f"0.3*sin(2*{const}*t) + E*0"}, "x+": {"derivative": 0.1}, "y": "periodic", "z": "periodic"})
        a = eq.solve(pde_hip.ScalarField(grid, y0), t_range=0.2, dt=0.01, solver="runge-kutta", backend="hip")
        eq2 = pde_hip.PDE({"c": f"0.4*laplace(c) + 0.1*cos({const}*t)"}, bc={"x": {"derivative": 0.1}, "y": "periodic", "z": "periodic"})
        b = eq2.solve(pde_hip.ScalarField(grid, y0), t_range=0.2, dt=0.01, solver="euler", backend="hip")
        runs.append((np.array(a.data), np.array(b.data)))
    assert np.isfinite(runs[0][0]).all() and np.abs(runs[0][0] - y0).max() > 1e-3
    np.testing.assert_array_equal(runs[0][0], runs[1][0])
    np.testing.assert_array_equal(runs[0][1], runs[1][1])


class _DeviceClippedDiffusion(pde_hip.DiffusionPDE):
    """A hook the tracer can express (pde_hip/hooks.py): clipping from above, a time-dependent floor - no auxiliary data."""

    def make_post_step_hook(self, state):
        def hook(state_data, t, post_step_data):
            state_data[state_data > 0.6] = 0.6
            np.maximum(state_data, 0.05 * np.tanh(t), out=state_data)
            return state_data, post_step_data

        return hook, None


@pytest.mark.parametrize("solver,adaptive", [("euler", False), ("runge-kutta", False), ("euler", True)])
@pytest.mark.parametrize("shape", [(24, 32), (6, 8, 130)])
def test_traced_post_step_hook_runs_on_the_device(rng, shape, solver, adaptive, monkeypatch):
    """VERDICT r3 "missing #4": a pointwise hook becomes ONE run-time compiled pass per step - no download / upload of the state per step
    (counted) - and gives the bits of the host round trip (PDEHIP_DEVICE_HOOKS=0: the hook on downloaded numpy arrays)."""
    from pde_hip import device

    grid = pde_hip.UnitGrid(shape, periodic=[True] + [False] * (len(shape) - 1))
    y0 = rng.uniform(0, 1, shape)
    eq = _DeviceClippedDiffusion(0.8, bc="auto_periodic_neumann")
    kw = dict(t_range=0.6, dt=0.05, solver=solver, adaptive=adaptive, backend="hip", ret_info=True)
    counts = {"down": 0}
    orig = device.DeviceArray.get_valid

    def counting(self, *a, **k):
        counts["down"] += 1
        return orig(self, *a, **k)

    monkeypatch.setattr(device.DeviceArray, "get_valid", counting)
    monkeypatch.setenv("PDEHIP_DEVICE_HOOKS", "1")      # (a class with its own make_post_step_hook is traced on request only)
    res, info = eq.solve(pde_hip.ScalarField(grid, y0), **kw)
    on_device = np.array(res.data)
    downloads_device = counts["down"]
    # ... a hook given as PDE(..., post_step_hook=f) is traced by default
    monkeypatch.delenv("PDEHIP_DEVICE_HOOKS")
    counts["down"] = 0

    def clip(state_data, t):
        state_data[state_data > 0.6] = 0.6
        return state_data

    plain = pde_hip.PDE({"c": "0.8 * laplace(c)"}, post_step_hook=clip)
    res_plain = plain.solve(pde_hip.ScalarField(grid, y0), **kw)[0]
    assert counts["down"] <= 2 and np.array(res_plain.data).max() <= 0.6
    monkeypatch.setenv("PDEHIP_DEVICE_HOOKS", "0")
    counts["down"] = 0
    res_host, info_host = eq.solve(pde_hip.ScalarField(grid, y0), **kw)
    assert downloads_device <= 2 < counts["down"] and info["solver"]["steps"] == info_host["solver"]["steps"] >= 5
    np.testing.assert_array_equal(on_device, np.array(res_host.data))
    assert on_device.max() <= 0.6 and on_device.min() >= 0.0


def test_resident_state_and_numpy_views_held_by_the_caller(rng):
    """VERDICT r3 "weak #10": with the state resident on the device, a numpy VIEW of `state.data` taken before a stepper call is not updated
    by the call (the reference's steppers write through such views); any access to `field.data` / `field._data_full` afterwards brings
    the host copy - and thereby the old view, which aliases it - up to date.  `resident_state=False` gives the reference's behaviour at
    the price of a download per stepper call.  Both pinned here on the device."""
    from pde_hip.solvers import EulerSolver

    grid = pde_hip.UnitGrid([24, 16, 64], periodic=True)
    y0 = rng.uniform(0, 1, grid.shape)
    eq = pde_hip.DiffusionPDE(0.5)
    expect = eq.solve(pde_hip.ScalarField(grid, y0), t_range=0.5, dt=0.1, solver="euler", backend="hip").data

    state = pde_hip.ScalarField(grid, y0)
    view = state.data                       # taken BEFORE the run
    stepper = EulerSolver(eq, backend="hip").make_stepper(state, 0.1)
    stepper(state, 0.0, 0.5)
    np.testing.assert_array_equal(view, y0)                 # stale: the device advanced, the host copy was not touched
    np.testing.assert_array_equal(state.data, expect)       # an access to the field downloads ...
    np.testing.assert_array_equal(view, expect)             # ... into the same memory the old view looks at

    backend = pde_hip.HipBackend(config={"resident_state": False})
    state2 = pde_hip.ScalarField(grid, y0)
    view2 = state2.data
    stepper2 = EulerSolver(eq, backend=backend).make_stepper(state2, 0.1)
    stepper2(state2, 0.0, 0.5)
    np.testing.assert_array_equal(view2, expect)            # written through, like the reference (pde/backends/torch/backend.py:654-662)
