"""BASELINE config 5 (PDE({'c': 'laplace(c**3 - c - laplace(c))'}) on 256^3 fp32, adaptive RKF45) through eq.solve: microseconds per attempt
by differential timing of two run lengths (the measurement of bench.py's `extra`).  usage: python tools/time_cfg5.py [n=256] [dtype=float32]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import numpy as np

import pde_hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dtype = np.dtype(sys.argv[2]) if len(sys.argv) > 2 else np.dtype("float32")
backend = pde_hip.get_backend("hip")
grid = pde_hip.UnitGrid([n] * 3, periodic=True)
eq = pde_hip.PDE({"c": "laplace(c**3 - c - laplace(c))"})
state = pde_hip.ScalarField(grid, np.random.default_rng(0).uniform(-0.1, 0.1, grid.shape), dtype=dtype)
kw = dict(dt=1e-3, solver="runge-kutta", backend=backend, adaptive=True, ret_info=True)
eq.solve(state, t_range=1.0, **kw)
backend.synchronize()
best = None
for _ in range(3):
    t0 = time.perf_counter(); _, short = eq.solve(state, t_range=0.02, **kw); backend.synchronize(); w_short = time.perf_counter() - t0
    t0 = time.perf_counter(); _, info = eq.solve(state, t_range=1.0, **kw); backend.synchronize(); w = time.perf_counter() - t0
    a, a_short = info["solver"]["attempts"], short["solver"]["attempts"]
    per = (w - w_short) / (a - a_short) * 1e6
    best = per if best is None else min(best, per)
print(f"CFG5 n={n} {dtype}: {best:.1f} us per attempt ({info['solver']['steps']} steps, {a} attempts)")
