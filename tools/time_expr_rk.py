"""RK4 steps / RKF45 attempts of an expression PDE through the Python-level stepper, stages as one sweep vs separate
kernels (STAGE=0).  usage: [STAGE=0] python tools/time_expr_rk.py [size]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "py-pde_amd"))
import pde_hip  # noqa: E402
from pde_hip.expr import ExpressionRhs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
if os.environ.get("STAGE", "1") == "0":
    ExpressionRhs._stage_ok = False
b = pde_hip.get_backend("hip")
grid = pde_hip.UnitGrid([n] * 3, periodic=True)
state = pde_hip.ScalarField(grid, np.random.default_rng(0).uniform(-0.1, 0.1, grid.shape))
for name, eq in [("Allen-Cahn", pde_hip.PDE({"c": "c - c**3 + laplace(c)"})), ("KPZ", pde_hip.PDE({"h": "0.7*laplace(h) + 0.3*gradient_squared(h)"}))]:
    for label, kw in [("RK4 fixed dt", dict(dt=1e-3, adaptive=False)), ("RKF45 adaptive", dict(dt=None, adaptive=True, tolerance=1e-6))]:
        solver = pde_hip.solvers.SolverBase.from_name("runge-kutta", pde=eq, backend=b, **{k: v for k, v in kw.items() if k != "dt"})
        solver.info.update(dt=kw["dt"] if kw["dt"] else 1e-3, steps=0)
        stepper = b.make_inner_stepper(solver, state)
        from pde_hip.device import DeviceArray
        arr = DeviceArray(b.grid_info(grid, state.dtype)).set_valid(state.data)
        stepper(arr, 0.0, 4e-3)   # warm-up (JIT, allocations)
        b.synchronize()
        s0 = solver.info["steps"]
        lib = b._lib
        e0, e1, ms = C.c_void_p(), C.c_void_p(), C.c_float()
        lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
        t0 = time.perf_counter()
        lib.event_record(e0, b.stream)
        stepper(arr, 0.0, 60e-3)
        t_enq = time.perf_counter() - t0
        lib.event_record(e1, b.stream)
        b.synchronize()
        wall = time.perf_counter() - t0
        lib.event_elapsed_ms(e0, e1, C.byref(ms))
        steps = solver.info["steps"] - s0
        print(f"| {n}^3 {name} | {label} | stage sweeps={os.environ.get('STAGE', '1')} | {steps} steps | {wall / steps * 1e3:.3f} ms/step wall | "
              f"{ms.value / steps:.3f} ms/step device | host enqueue {t_enq / steps * 1e3:.3f} ms/step |", flush=True)
