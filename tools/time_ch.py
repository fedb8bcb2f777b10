"""Time Cahn-Hilliard steps (fused one-sweep kernel when covered): Euler via pdehip_euler_run, and one RKF45 attempt.

usage: [PDEHIP_EULER2=off] python tools/time_ch.py [size] [steps] [dtype]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "py-pde_amd"))
import pde_hip  # noqa: E402
from pde_hip.device import DeviceArray, DeviceScalar, ptr_array  # noqa: E402

shape = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256").split(",")]
if len(shape) == 1:
    shape = shape * 3
n = "x".join(map(str, shape))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dtype = np.dtype(sys.argv[3]) if len(sys.argv) > 3 else np.dtype("float64")
backend = pde_hip.get_backend("hip")
lib = backend._lib
grid = pde_hip.UnitGrid(shape, periodic=True)
state = pde_hip.ScalarField.random_uniform(grid, -0.1, 0.1, rng=np.random.default_rng(0), dtype=dtype)
spec = backend.make_rhs_spec(pde_hip.CahnHilliardPDE(1.0), state)
a, b = DeviceArray(spec.info).set_valid(state.data), DeviceArray(spec.info)
stream = C.c_void_p()
lib.stream_create(C.byref(stream))
e0, e1 = C.c_void_p(), C.c_void_p()
lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
res = C.c_void_p()
ms = C.c_float()
lib.euler_run(spec.info.ref, spec.ref, a.ptr, b.ptr, 1e-3, 10, C.byref(res), stream)
lib.stream_synchronize(stream)
best = 1e9
for _ in range(3):
    lib.event_record(e0, stream)
    lib.euler_run(spec.info.ref, spec.ref, a.ptr, b.ptr, 1e-3, steps, C.byref(res), stream)
    lib.event_record(e1, stream)
    lib.stream_synchronize(stream)
    lib.event_elapsed_ms(e0, e1, C.byref(ms))
    best = min(best, ms.value / steps)
cells = int(np.prod(shape))
tag = f"EULER2={os.environ.get('PDEHIP_EULER2', 'default'):>8s} CH n={n} {dtype}"
print(f"{tag}: Euler {best:.4f} ms/step  {cells / best / 1e6:.1f} Gcells/s")
work = [DeviceArray(spec.info) for _ in range(7)]
ynew, err = DeviceArray(spec.info), DeviceScalar()
best = 1e9
for _ in range(4):
    lib.event_record(e0, stream)
    for _ in range(10):
        lib.rkf45_attempt(spec.info.ref, spec.ref, a.ptr, ynew.ptr, ptr_array(work), 1e-3, err.ptr, stream)
    lib.event_record(e1, stream)
    lib.stream_synchronize(stream)
    lib.event_elapsed_ms(e0, e1, C.byref(ms))
    best = min(best, ms.value / 10)
print(f"{tag}: RKF45 attempt {best:.4f} ms  {cells / best / 1e6:.1f} Gcells/s")
