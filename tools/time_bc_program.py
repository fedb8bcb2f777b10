"""Cost of conditions that are refreshed inside the C loops: ms per Euler step of 3-D fp64 diffusion with (a) constant
conditions on all six faces (two steps per sweep), (b) conditions that depend on time and position on all six faces, (c) two of them
reading the field as well (one step per sweep + one refresh launch per step).  `python tools/time_bc_program.py [n] [steps]`."""

from __future__ import annotations

import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "py-pde_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))
import pde_hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
grid = pde_hip.CartesianGrid([[0, 1]] * 3, [n] * 3, periodic=False)
dt = 0.1 * float(grid.discretization[0]) ** 2
data = np.random.default_rng(0).uniform(-1, 1, grid.shape)
CASES = {
    "constant": {"x-": {"value": 0.2}, "x+": {"derivative": 0.1}, "y-": {"value": -0.1}, "y+": {"derivative": 0}, "z-": {"value": 0.3}, "z+": {"derivative": -0.2}},
    "time_dependent": {"x-": {"value_expression": "0.2 * sin(3 * t) + 0.05 * y"}, "x+": {"derivative_expression": "0.1 * cos(t) * z"},
                       "y-": {"value_expression": "x * z * (1 + t)"}, "y+": {"derivative_expression": "0.05 * x * sin(t)"},
                       "z-": {"value_expression": "tanh(x - y) * t"}, "z+": {"derivative_expression": "0.1 * y * cos(2 * t)"}},
    "reads_the_field": {"x-": {"derivative_expression": "-0.3 * value**3 + 0.05 * y"}, "x+": {"value_expression": "0.2 * tanh(value) + 0.1 * sin(t)"},
                        "y-": {"value_expression": "x * z * (1 + t)"}, "y+": {"derivative_expression": "0.05 * x * sin(t)"},
                        "z-": {"value_expression": "tanh(x - y) * t"}, "z+": {"derivative_expression": "0.1 * y * cos(2 * t)"}},
}
out = {"n": n, "steps": steps}
for name, bc in CASES.items():
    eq = pde_hip.DiffusionPDE(1.0, bc=bc)
    state = pde_hip.ScalarField(grid, data)
    eq.solve(state, 4 * dt, dt, solver="euler")              # build kernels / programs
    def best_of(count, reps=3):
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            res = eq.solve(state, count * dt, dt, solver="euler")
            float(res.data[0, 0, 0])                         # the result on the host: the loop has finished
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        return best, res

    t1, res = best_of(steps)
    t2, res = best_of(3 * steps)
    # the loop alone: the difference of two run lengths cancels upload, download and the run-time compilation of the program
    out[name] = {"ms_per_step": round((t2 - t1) / (2 * steps) * 1e3, 4), "ms_per_step_incl_setup_and_transfers": round(t1 / steps * 1e3, 4)}
    assert np.isfinite(res.data).all()
print("BCPROG " + json.dumps(out))
