"""BASELINE cfg5 (256^3 fp32 `PDE` RKF45) under cProfile; run it under `rocprofv3 --kernel-trace --stats` for the kernel split."""
import cProfile, pstats, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import numpy as np
import pde_hip
rng = np.random.default_rng(0)
grid = pde_hip.UnitGrid([256] * 3, periodic=True)
eq = pde_hip.PDE({"c": "laplace(c**3 - c - laplace(c))"})
state = pde_hip.ScalarField(grid, rng.uniform(-0.1, 0.1, grid.shape), dtype=np.float32)
b = pde_hip.get_backend("hip")
eq.solve(state, t_range=0.02, dt=None, solver="runge-kutta", backend=b)
b.synchronize()
pr = cProfile.Profile()
pr.enable()
res, info = eq.solve(state, t_range=1.0, dt=None, solver="runge-kutta", backend=b, ret_info=True)
b.synchronize()
pr.disable()
print(info["solver"]["steps"], info["controller"]["profiler"])
pstats.Stats(pr).sort_stats("cumtime").print_stats(28)
