"""Generate the golden vectors under tests/golden/ by running the REFERENCE (py-pde).

Run in the build container only (needs /root/reference; the GPU box has no reference):

    python tests/golden/make_golden.py

What is recorded (all seeds fixed, tiny grids like the reference's own operator tests,
``tests/backends/generic/operators/test_cartesian_operators.py:20``):

* ``ops.npz`` — for a list of (grid, bc) cases: the input field, the full array after the
  reference's pure-numpy ``set_ghost_cells`` (``pde/grids/boundaries/axes.py:458-474``), and the
  operator results of the reference's eager torch-CPU backend (``config["backend.torch.compile"] =
  False``; expression order identical to the numba source) and of its scipy backend
  (the reference's own cross-backend yardstick) where that backend defines the operator.
* ``steppers.npz`` — final states / step counts of ``eq.solve`` for DiffusionPDE and
  CahnHilliardPDE with Euler (fixed + adaptive) and Runge–Kutta (RK4 fixed + RKF45 adaptive):
  torch-CPU backend for fixed Euler, numpy backend + scipy operators for the rest (the only
  reference path that runs RK without numba).

The case definitions (JSON) are stored inside the npz files, so the tests rebuild grid and BCs
through the package's own parser — which pins the parser against the reference as well.
"""

from __future__ import annotations

import json
import sys
import warnings
from pathlib import Path

import numpy as np

sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")

import pde  # noqa: E402
from pde import config  # noqa: E402

config["backend.torch.compile"] = False
HERE = Path(__file__).resolve().parent

OP_CASES = [
    # id, bounds, shape, periodic, bc (JSON-able), dtype
    dict(id="1d_periodic", bounds=[[0, 3.0]], shape=[8], periodic=[True], bc="auto_periodic_neumann"),
    dict(id="1d_dirichlet_neumann", bounds=[[-1, 1.5]], shape=[7], periodic=[False], bc={"x-": {"value": 1.5}, "x+": {"derivative": -0.3}}),
    dict(id="1d_mixed_curv", bounds=[[0, 2.0]], shape=[5], periodic=[False], bc={"x-": {"type": "mixed", "value": 2.0, "const": 0.7}, "x+": {"curvature": 0.4}}),
    dict(id="1d_single_cell", bounds=[[0, 1.0]], shape=[1], periodic=[True], bc="periodic"),
    dict(id="2d_unit_neumann", bounds=[[0, 6], [0, 4]], shape=[6, 4], periodic=[False, False], bc="auto_periodic_neumann"),
    dict(id="2d_aniso_mixed_bcs", bounds=[[0, 3], [1, 4.5]], shape=[6, 5], periodic=[False, True], bc={"x-": {"value": 1.5}, "x+": {"derivative": 0.3}, "y": "periodic"}),
    dict(id="2d_antiperiodic", bounds=[[0, 2], [0, 2]], shape=[4, 4], periodic=[True, True], bc={"x": "anti-periodic", "y": "periodic"}),
    dict(id="2d_inhomogeneous", bounds=[[0, 1], [0, 2]], shape=[4, 6], periodic=[False, False], bc={"x-": {"value": "sin(y)"}, "x+": {"derivative": [0.1, 0.2, 0.3, 0.4, 0.5, 0.6]}, "y-": "extrapolate", "y+": {"type": "mixed", "value": [1.0, 0.5, 2.0, 0.0], "const": [0.1, 0.2, 0.3, 0.4]}}),
    dict(id="2d_singular_x", bounds=[[0, 1], [0, 4]], shape=[1, 4], periodic=[True, False], bc="auto_periodic_neumann"),
    dict(id="2d_even_16", bounds=[[0, 16], [0, 16]], shape=[16, 16], periodic=[True, True], bc="auto_periodic_neumann"),
    dict(id="3d_periodic", bounds=[[0, 2], [0, 3], [0, 4]], shape=[4, 6, 8], periodic=[True, True, True], bc="auto_periodic_neumann"),
    dict(id="3d_mixed_bcs", bounds=[[0, 2.5], [0, 3], [-1, 1]], shape=[5, 4, 6], periodic=[False, True, False], bc={"x-": {"value": 0.5}, "x+": {"derivative": -1.0}, "y": "periodic", "z-": {"curvature": 0.2}, "z+": {"type": "mixed", "value": 0.5, "const": 1.0}}),
    dict(id="3d_singular", bounds=[[0, 1], [0, 1], [0, 4]], shape=[1, 1, 4], periodic=[True, False, False], bc="auto_periodic_neumann"),
    dict(id="3d_odd_fast_axis", bounds=[[0, 3], [0, 3], [0, 5]], shape=[3, 3, 5], periodic=[False, False, False], bc={"value": 0.25}),
    dict(id="2d_f32", bounds=[[0, 8], [0, 8]], shape=[8, 8], periodic=[True, False], bc="auto_periodic_neumann", dtype="float32"),
    dict(id="3d_f32", bounds=[[0, 4], [0, 4], [0, 8]], shape=[4, 4, 8], periodic=[True, True, True], bc="auto_periodic_neumann", dtype="float32"),
]

VECTOR_BC = "auto_periodic_neumann"


def make_grid(case):
    return pde.CartesianGrid(case["bounds"], case["shape"], periodic=case["periodic"])


def gen_ops():
    rng = np.random.default_rng(0)
    out = {"cases": json.dumps(OP_CASES)}
    for case in OP_CASES:
        cid = case["id"]
        dtype = np.dtype(case.get("dtype", "float64"))
        grid = make_grid(case)
        field = pde.ScalarField.random_uniform(grid, -1, 1, rng=rng, dtype=dtype)
        out[f"{cid}/input"] = field.data.copy()
        # ghost cells by the reference's numpy implementation
        f2 = field.copy()
        f2.set_ghost_cells(case["bc"])
        out[f"{cid}/full"] = f2._data_full.copy()
        iso = np.allclose(grid.discretization, grid.discretization[0])
        out[f"{cid}/laplace_torch"] = field.laplace(case["bc"], backend="torch").data
        if iso and dtype == np.float64:
            out[f"{cid}/laplace_scipy"] = field.laplace(case["bc"], backend="scipy").data
        out[f"{cid}/gradient_central_torch"] = field.gradient(case["bc"], backend="torch").data
        if dtype == np.float64:
            for method in ["central", "forward", "backward"]:
                out[f"{cid}/gradient_{method}_scipy"] = field.gradient(case["bc"], backend="scipy", method=method).data
        out[f"{cid}/gradient_squared_central_torch"] = field.apply_operator("gradient_squared", case["bc"], backend="torch").data
        try:
            out[f"{cid}/gradient_squared_noncentral_torch"] = field.apply_operator("gradient_squared", case["bc"], backend="torch", central=False).data
        except Exception as err:  # noqa: BLE001
            print(f"[{cid}] gradient_squared(central=False) unavailable in torch backend: {err}")
        # vector field operators with the default BCs
        vec = pde.VectorField.random_uniform(grid, -1, 1, rng=rng, dtype=dtype)
        out[f"{cid}/vector_input"] = vec.data.copy()
        v2 = vec.copy()
        v2.set_ghost_cells(VECTOR_BC)
        out[f"{cid}/vector_full"] = v2._data_full.copy()
        out[f"{cid}/divergence_central_torch"] = vec.divergence(VECTOR_BC, backend="torch").data
        if dtype == np.float64:
            for method in ["central", "forward", "backward"]:
                out[f"{cid}/divergence_{method}_scipy"] = vec.divergence(VECTOR_BC, backend="scipy", method=method).data
        out[f"{cid}/vector_laplace_torch"] = vec.laplace(VECTOR_BC, backend="torch").data
        out[f"{cid}/vector_gradient_torch"] = vec.gradient(VECTOR_BC, backend="torch").data
    # a vector field with a normal BC
    grid = pde.UnitGrid([4, 5], periodic=[False, True])
    vec = pde.VectorField.random_uniform(grid, -1, 1, rng=rng)
    bc = {"x": {"normal_value": 0.5}, "y": "periodic"}
    v2 = vec.copy()
    v2.set_ghost_cells(bc)
    out["normal_bc/input"] = vec.data.copy()
    out["normal_bc/full"] = v2._data_full.copy()
    out["normal_bc/bc"] = json.dumps(bc)
    np.savez_compressed(HERE / "ops.npz", **out)
    print("ops.npz:", len(out), "arrays")


STEP_CASES = [
    dict(id="diff2d_euler_torch", pde="diffusion", D=0.7, bounds=[[0, 3], [1, 4.5]], shape=[6, 5], periodic=[False, True],
         bc={"x-": {"value": 1.5}, "x+": {"derivative": 0.3}, "y": "periodic"}, solver="euler", dt=0.01, t_range=1.0, backend="torch"),
    dict(id="diff3d_euler_torch", pde="diffusion", D=1.0, bounds=[[0, 4], [0, 6], [0, 8]], shape=[4, 6, 8], periodic=[True, True, True],
         bc="auto_periodic_neumann", solver="euler", dt=0.1, t_range=5.0, backend="torch"),
    dict(id="diff64_euler_torch", pde="diffusion", D=1.0, bounds=[[0, 64], [0, 64]], shape=[64, 64], periodic=[False, False],
         bc="auto_periodic_neumann", solver="euler", dt=0.1, t_range=10.0, backend="torch", vmin=0, vmax=1),
    dict(id="ch2d_euler_torch", pde="cahn_hilliard", gamma=1.0, bounds=[[0, 8], [0, 8]], shape=[8, 8], periodic=[True, False],
         bc="auto_periodic_neumann", solver="euler", dt=1e-3, t_range=0.2, backend="torch"),
    dict(id="ch2d_euler_torch_gamma", pde="cahn_hilliard", gamma=0.6, bounds=[[0, 16], [0, 16]], shape=[16, 16], periodic=[True, True],
         bc="auto_periodic_neumann", solver="euler", dt=1e-3, t_range=0.1, backend="torch"),
    dict(id="diff2d_rk4_numpy", pde="diffusion", D=0.5, bounds=[[0, 8], [0, 8]], shape=[8, 8], periodic=[False, True],
         bc="auto_periodic_neumann", solver="runge-kutta", dt=0.05, t_range=1.0, backend="numpy"),
    dict(id="ch2d_rk4_numpy", pde="cahn_hilliard", gamma=1.0, bounds=[[0, 8], [0, 8]], shape=[8, 8], periodic=[True, False],
         bc="auto_periodic_neumann", solver="runge-kutta", dt=1e-3, t_range=0.05, backend="numpy"),
    dict(id="ch2d_rkf45_numpy", pde="cahn_hilliard", gamma=1.0, bounds=[[0, 8], [0, 8]], shape=[8, 8], periodic=[True, False],
         bc="auto_periodic_neumann", solver="runge-kutta", dt=None, t_range=0.5, backend="numpy"),
    dict(id="ch3d_rkf45_numpy", pde="cahn_hilliard", gamma=1.0, bounds=[[0, 8], [0, 8], [0, 8]], shape=[8, 8, 8], periodic=[True, True, True],
         bc="auto_periodic_neumann", solver="runge-kutta", dt=None, t_range=0.2, backend="numpy", vmin=-0.1, vmax=0.1),
    dict(id="diff2d_rkf45_numpy", pde="diffusion", D=1.0, bounds=[[0, 8], [0, 8]], shape=[8, 8], periodic=[False, False],
         bc={"value": 0.5}, solver="runge-kutta", dt=None, t_range=2.0, backend="numpy"),
    dict(id="diff2d_euler_adaptive_numpy", pde="diffusion", D=1.0, bounds=[[0, 8], [0, 8]], shape=[8, 8], periodic=[True, False],
         bc="auto_periodic_neumann", solver="euler", dt=None, t_range=1.0, backend="numpy"),
    dict(id="ch2d_f32_euler_torch", pde="cahn_hilliard", gamma=1.0, bounds=[[0, 8], [0, 8]], shape=[8, 8], periodic=[True, True],
         bc="auto_periodic_neumann", solver="euler", dt=1e-3, t_range=0.1, backend="torch", dtype="float32"),
    # Adams-Bashforth (pde/solvers/adams_bashforth.py; only the numpy and numba backends implement it)
    dict(id="diff2d_ab_numpy", pde="diffusion", D=0.5, bounds=[[0, 8], [0, 8]], shape=[8, 8], periodic=[False, True],
         bc="auto_periodic_neumann", solver="adams-bashforth", dt=0.05, t_range=1.0, backend="numpy"),
    dict(id="diff3d_ab_numpy", pde="diffusion", D=1.0, bounds=[[0, 4], [0, 6], [0, 8]], shape=[4, 6, 8], periodic=[True, False, True],
         bc={"x": "periodic", "y": {"value": 0.3}, "z": "periodic"}, solver="adams-bashforth", dt=0.05, t_range=0.55, backend="numpy"),
    dict(id="ch2d_ab_numpy", pde="cahn_hilliard", gamma=1.0, bounds=[[0, 8], [0, 8]], shape=[8, 8], periodic=[True, False],
         bc="auto_periodic_neumann", solver="adams-bashforth", dt=1e-3, t_range=0.05, backend="numpy"),
]


def gen_steppers():
    rng = np.random.default_rng(1)
    out = {"cases": json.dumps(STEP_CASES)}
    for case in STEP_CASES:
        cid = case["id"]
        grid = make_grid(case)
        dtype = np.dtype(case.get("dtype", "float64"))
        state = pde.ScalarField.random_uniform(grid, case.get("vmin", -0.5), case.get("vmax", 0.5), rng=rng, dtype=dtype)
        if case["pde"] == "diffusion":
            eq = pde.DiffusionPDE(case["D"], bc=case["bc"])
        else:
            eq = pde.CahnHilliardPDE(case["gamma"], bc_c=case["bc"], bc_mu=case["bc"])
        # the numpy backend routes operators through the default backend -> use scipy's
        config["default_backend"] = "scipy" if case["backend"] == "numpy" else "numba"
        res, info = eq.solve(state, t_range=case["t_range"], dt=case["dt"], backend=case["backend"], solver=case["solver"],
                             tracker=None, ret_info=True)
        out[f"{cid}/input"] = state.data.copy()
        out[f"{cid}/final"] = res.data.copy()
        sinfo = info["solver"]
        out[f"{cid}/steps"] = np.array(sinfo["steps"])
        out[f"{cid}/dt_last"] = np.array(sinfo["dt"], dtype=np.float64)
        out[f"{cid}/t_final"] = np.array(info["controller"]["t_final"], dtype=np.float64)
        print(f"[{cid}] steps={sinfo['steps']} dt_last={sinfo['dt']} t_final={info['controller']['t_final']}")
    config["default_backend"] = "numba"
    np.savez_compressed(HERE / "steppers.npz", **out)
    print("steppers.npz:", len(out), "arrays")


if __name__ == "__main__":
    gen_ops()
    gen_steppers()
