"""Timings of run-time specialised expression kernels (one Euler update = all passes of the expression)."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import numpy as np

import pde_hip
from pde_hip.device import DeviceArray

b = pde_hip.get_backend("hip")
lib = b._lib
ev = [C.c_void_p() for _ in range(2)]
for e in ev:
    lib.event_create(C.byref(e))
ms = C.c_float()
print("| grid | dtype | expression | passes | ms / Euler update | Gcells/s | algorithmic GB/s (2 values per pass) | % of 8 TB/s |")
print("|---|---|---|---:|---:|---:|---:|---:|")
for shape, dtype in [((512, 512, 512), np.float64), ((4096, 4096), np.float64), ((256, 256, 256), np.float32)]:
    grid = pde_hip.UnitGrid(shape, periodic=True)
    state = pde_hip.ScalarField(grid, np.random.default_rng(0).uniform(-0.5, 0.5, shape), dtype=dtype)
    cells = int(np.prod(shape))
    it = np.dtype(dtype).itemsize
    for name, rhs, consts in [("Allen-Cahn c - c**3 + laplace(c)", {"c": "c - c**3 + laplace(c)"}, {}),
                              ("KPZ nu*laplace(h) + lam*gradient_squared(h)", {"h": "nu*laplace(h) + lam*gradient_squared(h)"}, {"nu": 0.5, "lam": 1.5}),
                              ("Swift-Hohenberg (2 passes)", {"c": "(eps - 1)*c - 2*laplace(c) - laplace(laplace(c)) - c**3"}, {"eps": 0.1}),
                              ("Cahn-Hilliard via expression compiler (2 passes)", {"c": "laplace(c**3 - c - laplace(c)) + 0*c"}, {})]:
        eq = pde_hip.PDE(rhs, consts=consts)
        erhs = b.make_expression_rhs(eq, state)
        y, out = DeviceArray(erhs.info).set_valid(state.data), DeviceArray(erhs.info)
        for _ in range(3):
            erhs.apply(y, out, "euler", 1e-3, 0.0)
        lib.stream_synchronize(None)
        reps = 30
        lib.event_record(ev[0], None)
        for _ in range(reps):
            erhs.apply(y, out, "euler", 1e-3, 0.0)
        lib.event_record(ev[1], None)
        lib.stream_synchronize(None)
        lib.event_elapsed_ms(ev[0], ev[1], C.byref(ms))
        t = ms.value / reps
        npass = len(erhs.plan.passes)
        extra_reads = sum(len(p.extras) for p in erhs.plan.passes) + (1 if erhs.plan.passes[-1].src != "state" else 0)
        bpc = (2 * npass + extra_reads) * it
        print(f"| {'x'.join(map(str, shape))} | {np.dtype(dtype).name} | {name} | {npass} | {t:.4f} | {cells/t/1e6:.1f} | {cells*bpc/t/1e6:.0f} | {cells*bpc/t/1e6/80:.1f} |", flush=True)
        # two Euler steps per sweep (one-pass expressions only): ms per STEP
        if erhs.euler2(y, out, 1e-3):
            lib.stream_synchronize(None)
            lib.event_record(ev[0], None)
            for _ in range(reps):
                erhs.euler2(y, out, 1e-3)
            lib.event_record(ev[1], None)
            lib.stream_synchronize(None)
            lib.event_elapsed_ms(ev[0], ev[1], C.byref(ms))
            t2 = ms.value / reps / 2
            print(f"| {'x'.join(map(str, shape))} | {np.dtype(dtype).name} | {name}, TWO steps per sweep | 1/2 | {t2:.4f} | {cells/t2/1e6:.1f} | {cells*2*it/t2/1e6:.0f} | {cells*2*it/t2/1e6/80:.1f} |", flush=True)
