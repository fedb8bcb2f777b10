"""Euler-step and Laplacian timings for sizes around the vector / tile boundaries (the "odd-size cliff" of VERDICT r1 #10).
usage: python tools/time_sizes.py [n ...]   (default: 510 511 512 513 and 500x500x300)
"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import numpy as np

import pde_hip
from pde_hip import _abi
from pde_hip.device import DeviceArray

b = pde_hip.get_backend("hip")
lib = b._lib
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(510,) * 3, (511,) * 3, (512,) * 3, (513,) * 3, (500, 500, 300), (511, 511, 512), (4095, 4097), (4096, 4096)]
print("| grid | dtype | laplace ms | laplace frac of 8 TB/s | Euler ms/step | Gcell-steps/s | effective frac |")
print("|---|---|---:|---:|---:|---:|---:|")
for shape in shapes:
    for dtype in (np.float64, np.float32):
        grid = pde_hip.UnitGrid(shape, periodic=True)
        eq = pde_hip.DiffusionPDE()
        state = pde_hip.ScalarField(grid, np.random.default_rng(0).random(shape), dtype=dtype)
        spec = b.make_rhs_spec(eq, state)
        info = spec.info
        a, bb = DeviceArray(info).set_valid(state.data), DeviceArray(info)
        lib.set_ghost_cells(info.ref, 1, spec.bc_c.c, a.ptr, None)
        ev = [C.c_void_p() for _ in range(2)]
        for e in ev:
            lib.event_create(C.byref(e))
        cells, w = int(np.prod(shape)), np.dtype(dtype).itemsize

        def timed(fn, reps):
            fn()
            lib.stream_synchronize(None)
            lib.event_record(ev[0], None)
            for _ in range(reps):
                fn()
            lib.event_record(ev[1], None)
            lib.stream_synchronize(None)
            ms = C.c_float()
            lib.event_elapsed_ms(ev[0], ev[1], C.byref(ms))
            return ms.value / reps

        t_lap = timed(lambda: lib.laplace(info.ref, a.ptr, bb.ptr, _abi.OUT_FULL, None), 20)
        res = C.c_void_p()
        steps = 40
        t_eu = timed(lambda: lib.euler_run(info.ref, spec.ref, a.ptr, bb.ptr, 0.1, steps, C.byref(res), None), 3) / steps
        print(f"| {'x'.join(map(str, shape))} | {np.dtype(dtype).name} | {t_lap:.4f} | {cells * 2 * w / t_lap / 1e6 / 8000:.3f} | {t_eu:.4f} | {cells / t_eu / 1e6:.1f} | "
              f"{cells * 2 * w / t_eu / 1e6 / 8000:.3f} |", flush=True)
