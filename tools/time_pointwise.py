"""Time the pointwise Runge-Kutta kernels (lincomb with 1..5 terms, rk4_combine, rkf45_combine, ab2_combine) at one grid size.

usage: python tools/time_pointwise.py [size]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "py-pde_amd"))
import pde_hip  # noqa: E402
from pde_hip.device import DeviceArray, DeviceScalar, GridInfo, ptr_array  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lib = pde_hip.get_backend("hip")._lib
grid = pde_hip.UnitGrid([n, n, n], periodic=True)
info = GridInfo(grid.shape, grid.discretization, np.float64)
arrs = [DeviceArray(info) for _ in range(9)]
e0, e1 = C.c_void_p(), C.c_void_p()
lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
ms = C.c_float()
cells = n ** 3


def timeit(fn, units, name, reps=20):
    fn()
    lib.stream_synchronize(None)
    lib.event_record(e0, None)
    for _ in range(reps):
        fn()
    lib.event_record(e1, None)
    lib.stream_synchronize(None)
    lib.event_elapsed_ms(e0, e1, C.byref(ms))
    t = ms.value / reps
    print(f"| {n}^3 | {name} | {t:.4f} | {units} | {cells * 8 * units / t / 1e6:.0f} | {cells * 8 * units / t / 1e6 / 80:.1f} |", flush=True)


print("| grid | kernel | ms | arrays moved | GB/s | % of 8 TB/s |")
print("|---|---|---:|---:|---:|---:|")
for k in range(1, 6):
    cf = (C.c_double * k)(*[0.1 * (j + 1) for j in range(k)])
    ks = ptr_array(arrs[2:2 + k])
    timeit(lambda: lib.lincomb(info.ref, 1, arrs[0].ptr, arrs[1].ptr, k, cf, ks, None), k + 2, f"lincomb, {k} terms")
timeit(lambda: lib.rk4_combine(info.ref, 1, arrs[0].ptr, arrs[1].ptr, arrs[2].ptr, arrs[3].ptr, arrs[4].ptr, None), 6, "rk4_combine")
err = DeviceScalar()
k6 = ptr_array(arrs[2:8])
timeit(lambda: lib.rkf45_combine(info.ref, 1, arrs[0].ptr, arrs[1].ptr, k6, err.ptr, None), 8, "rkf45_combine (+ error norm)")
timeit(lambda: lib.ab2_combine(info.ref, 1, arrs[0].ptr, arrs[1].ptr, arrs[2].ptr, 0.01, None), 4, "ab2_combine")
timeit(lambda: lib.max_abs_diff(info.ref, 1, arrs[0].ptr, arrs[1].ptr, err.ptr, None), 2, "max_abs_diff")
