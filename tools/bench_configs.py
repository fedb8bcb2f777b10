"""Single-GPU timings of the BASELINE.json configs through the mirror API (`eq.solve`), state device
resident inside one stepper call.  Prints a small table (markdown) — evidence for DESIGN.md / profiles/.
usage: [ONLY=cfg5] python tools/bench_configs.py
"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import numpy as np

import pde_hip

rng = np.random.default_rng(0)
ONLY = os.environ.get("ONLY", "")   # run only the configs whose name contains this string (profiling aid)


def run(name, eq, grid, dtype, t_range, dt, solver, lo=0.0, hi=1.0, **kw):
    if ONLY and ONLY not in name:
        return
    state = pde_hip.ScalarField(grid, rng.uniform(lo, hi, grid.shape), dtype=dtype)
    b = pde_hip.get_backend("hip")
    eq.solve(state, t_range=t_range / 50, dt=dt, solver=solver, backend=b, **kw)  # warm-up (allocations)
    b.synchronize()
    t0 = time.perf_counter()
    res, info = eq.solve(state, t_range=t_range, dt=dt, solver=solver, backend=b, ret_info=True, **kw)
    b.synchronize()
    wall = time.perf_counter() - t0               # upload + all steps on the device (the state stays resident: no download yet)
    data = res.data                                # first host access: ONE download
    wall_dl = time.perf_counter() - t0
    steps = info["solver"]["steps"]
    cells = int(np.prod(grid.shape))
    assert np.isfinite(data).all()
    print(f"| {name} | {steps} | {wall*1e3:.1f} | {wall/steps*1e6:.1f} | {cells*steps/wall/1e6:.0f} | {wall_dl*1e3:.1f} |", flush=True)


print("| config | steps | eq.solve wall until the device is done (ms; incl. state copy + upload) | us/step | Mcell-steps/s | ... incl. the download of the result (ms) |")
print("|---|---:|---:|---:|---:|---:|")
run("cfg1 Diffusion UnitGrid 64^2 fp64 Euler dt=0.1", pde_hip.DiffusionPDE(), pde_hip.UnitGrid([64, 64]), np.float64, 100.0, 0.1, "euler")
run("cfg2 Diffusion 1024^2 fp64 periodic Euler dt=0.1", pde_hip.DiffusionPDE(), pde_hip.CartesianGrid([[0, 1024]] * 2, 1024, periodic=True), np.float64, 100.0, 0.1, "euler")
run("cfg3 CahnHilliard UnitGrid 512^2 fp64 Euler dt=1e-3 (reference benchmark, 1e4 steps)", pde_hip.CahnHilliardPDE(), pde_hip.UnitGrid([512, 512]), np.float64, 10.0, 1e-3, "euler")
run("cfg3b CahnHilliard 512^2 RK4 dt=1e-2", pde_hip.CahnHilliardPDE(), pde_hip.UnitGrid([512, 512]), np.float64, 10.0, 1e-2, "runge-kutta")
run("cfg3c CahnHilliard 512^2 RKF45 adaptive", pde_hip.CahnHilliardPDE(), pde_hip.UnitGrid([512, 512]), np.float64, 10.0, None, "runge-kutta")
run("cfg4 Diffusion 512^3 fp64 periodic Euler dt=0.1 (1 GPU)", pde_hip.DiffusionPDE(), pde_hip.UnitGrid([512] * 3, periodic=True), np.float64, 20.0, 0.1, "euler")
run("cfg5 PDE laplace(c**3-c-laplace(c)) 256^3 fp32 RKF45 adaptive (1 GPU)", pde_hip.PDE({"c": "laplace(c**3 - c - laplace(c))"}),
    pde_hip.UnitGrid([256] * 3, periodic=True), np.float32, 1.0, None, "runge-kutta", lo=-0.1, hi=0.1)
