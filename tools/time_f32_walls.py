"""ms per Euler step of 3-D fp32 diffusion with constant conditions on all six faces (two steps per sweep): `python tools/time_f32_walls.py [n ...]`.
The difference of two run lengths cancels upload, download and setup."""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "py-pde_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))
import pde_hip  # noqa: E402

BC = {"x-": {"value": 0.2}, "x+": {"derivative": 0.1}, "y-": {"value": -0.1}, "y+": {"derivative": 0}, "z-": {"value": 0.3}, "z+": {"derivative": -0.2}}
for arg in sys.argv[1:] or ["512"]:
    shape = [int(v) for v in arg.split("x")] if "x" in arg else [int(arg)] * 3
    grid = pde_hip.CartesianGrid([[0, 1]] * 3, shape, periodic=False)
    dt = 0.1 * float(min(grid.discretization)) ** 2
    for dtype in (np.float32, np.float64):
        state = pde_hip.ScalarField(grid, np.random.default_rng(0).uniform(-1, 1, grid.shape).astype(dtype), dtype=dtype)
        eq = pde_hip.DiffusionPDE(1.0, bc=BC)
        eq.solve(state, 4 * dt, dt, solver="euler", backend="hip", tracker=None)

        def best_of(count, reps=3):
            best = None
            for _ in range(reps):
                t0 = time.perf_counter()
                res = eq.solve(state, count * dt, dt, solver="euler", backend="hip", tracker=None)
                float(res.data[0, 0, 0])
                el = time.perf_counter() - t0
                best = el if best is None else min(best, el)
            return best

        t1, t2 = best_of(200), best_of(600)
        ms = (t2 - t1) / 400 * 1e3
        print(f"WALLS {'x'.join(map(str, shape))} {np.dtype(dtype).name}: {ms:.4f} ms/step  {np.prod(shape) / ms * 1e-6:.1f} Gcell-steps/s", flush=True)
