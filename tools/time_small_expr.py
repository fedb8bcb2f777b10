"""Wall time per explicit Euler step of small 2-D expression PDEs through eq.solve (the common py-pde use): how much of it is
Python / launch overhead?  usage: python tools/time_small_expr.py [n]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import numpy as np

import pde_hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
solver = sys.argv[2] if len(sys.argv) > 2 else "euler"          # euler | runge-kutta | adaptive (Euler with error control)
grid = pde_hip.UnitGrid([n, n], periodic=True)
rng = np.random.default_rng(0)
c = pde_hip.ScalarField(grid, rng.uniform(-0.1, 0.1, grid.shape))
uv = pde_hip.FieldCollection([pde_hip.ScalarField(grid, rng.uniform(0.5, 1.5, grid.shape)), pde_hip.ScalarField(grid, rng.uniform(2.5, 3.5, grid.shape))])
cases = [
    ("DiffusionPDE (C loop, 8 steps per launch)", pde_hip.DiffusionPDE(), c, 1e-2),
    ("Allen-Cahn expression, 1 pass", pde_hip.PDE({"c": "c - c**3 + laplace(c)"}), c, 1e-2),
    ("Allen-Cahn + explicit time, 1 pass", pde_hip.PDE({"c": "c - c**3 + laplace(c) + 0.01*sin(t)"}), c, 1e-2),
    ("Swift-Hohenberg expression, 2 passes", pde_hip.PDE({"c": "(0.1 - 1) * c - 2 * laplace(c) - laplace(laplace(c)) - c**3"}), c, 1e-3),
    ("Brusselator, 2 fields", pde_hip.PDE({"u": "laplace(u) + 1 - 4 * u + v * u**2", "v": "0.1 * laplace(v) + 3 * u - v * u**2"}), uv, 1e-3),
    ("Burgers-type, d_dx", pde_hip.PDE({"c": "-c * d_dx(c) + 0.1 * laplace(c)"}), c, 1e-3),
]
steps = 4000
print(f"| {n}^2 fp64, {solver}, t_range = {steps} dt through eq.solve | us per step | steps |")
print("|---|---:|---:|")
for name, eq, state, dt in cases:
    kw = {"solver": "euler", "dt": None, "adaptive": True} if solver == "adaptive" else {"solver": solver, "dt": dt}
    eq.solve(state, t_range=20 * dt, backend="hip", **kw)   # builds / compiles
    pde_hip.get_backend("hip").synchronize()
    t0 = time.perf_counter()
    res, info = eq.solve(state, t_range=steps * dt, backend="hip", ret_info=True, **kw)
    _ = res.data.sum()
    t = time.perf_counter() - t0
    done = info["solver"]["steps"]
    print(f"| {name} | {t / done * 1e6:.1f} | {done} |", flush=True)
