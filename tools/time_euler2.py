"""Time pdehip_euler_run (two-steps-per-sweep kernel when eligible) for one PDEHIP_EULER2 setting.

usage: PDEHIP_EULER2="ry,blocks" python tools/time_euler2.py [size] [steps] [dtype]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "py-pde_amd"))
import pde_hip  # noqa: E402
from pde_hip.device import DeviceArray  # noqa: E402

shape = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512").split(",")]
if len(shape) == 1:
    shape = shape * 3
n = "x".join(map(str, shape))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dtype = np.dtype(sys.argv[3]) if len(sys.argv) > 3 else np.dtype("float64")
per = os.environ.get("TIME_PERIODIC", "1") == "1"
backend = pde_hip.get_backend("hip")
lib = backend._lib
grid = pde_hip.UnitGrid(shape, periodic=per)
state = pde_hip.ScalarField.random_uniform(grid, rng=np.random.default_rng(0), dtype=dtype)
spec = backend.make_rhs_spec(pde_hip.DiffusionPDE(1.0), state)
a, b = DeviceArray(spec.info).set_valid(state.data), DeviceArray(spec.info)
stream = C.c_void_p()
lib.stream_create(C.byref(stream))
e0, e1 = C.c_void_p(), C.c_void_p()
lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
res = C.c_void_p()
lib.euler_run(spec.info.ref, spec.ref, a.ptr, b.ptr, 0.1, 20, C.byref(res), stream)
lib.stream_synchronize(stream)
best = 1e9
for _ in range(3):
    lib.event_record(e0, stream)
    lib.euler_run(spec.info.ref, spec.ref, a.ptr, b.ptr, 0.1, steps, C.byref(res), stream)
    lib.event_record(e1, stream)
    lib.stream_synchronize(stream)
    ms = C.c_float()
    lib.event_elapsed_ms(e0, e1, C.byref(ms))
    best = min(best, ms.value / steps)
cells = int(np.prod(shape))
print(f"EULER2={os.environ.get('PDEHIP_EULER2', 'default'):>10s} n={n} {dtype} periodic={per}: {best:.4f} ms/step  {cells / best / 1e6:.1f} Gcells/s  "
      f"{cells * 2 * dtype.itemsize / best / 1e9 / 8:.1%} of 8 TB/s (16 B/cell-step)")
