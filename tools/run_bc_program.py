"""A short Euler run with conditions of time and position on all six faces of a 512^3 grid (the two-step sweep + shell kernel + refresh launch):
the workload of the counter passes of tools/gpu_r4_call17.sh.  `python tools/run_bc_program.py [n] [steps]`."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import pde_hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
grid = pde_hip.CartesianGrid([[0, 1]] * 3, [n] * 3, periodic=False)
dt = 0.1 * float(grid.discretization[0]) ** 2
bc = {"x-": {"value_expression": "0.2 * sin(3 * t) + 0.05 * y"}, "x+": {"derivative_expression": "0.1 * cos(t) * z"},
      "y-": {"value_expression": "x * z * (1 + t)"}, "y+": {"derivative_expression": "0.05 * x * sin(t)"},
      "z-": {"value_expression": "tanh(x - y) * t"}, "z+": {"derivative_expression": "0.1 * y * cos(2 * t)"}}
state = pde_hip.ScalarField(grid, np.random.default_rng(0).uniform(-1, 1, grid.shape))
res = pde_hip.DiffusionPDE(1.0, bc=bc).solve(state, steps * dt, dt, solver="euler")
print("done", float(res.data[0, 0, 0]))
