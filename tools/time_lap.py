"""Time the 3-D Laplacian (pdehip_laplace on a resident field, ghost cells set once) and the Euler run: one line each.
usage: [PDEHIP_LIB=...] python tools/time_lap.py [n=512] [dtype=float64]"""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import numpy as np

import pde_hip
from pde_hip import _abi
from pde_hip.device import DeviceArray

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dtype = np.dtype(sys.argv[2]) if len(sys.argv) > 2 else np.dtype("float64")
b = pde_hip.get_backend("hip")
lib = b._lib
grid = pde_hip.UnitGrid([n] * 3, periodic=True)
state = pde_hip.ScalarField(grid, np.random.default_rng(0).random((n,) * 3), dtype=dtype)
spec = b.make_rhs_spec(pde_hip.DiffusionPDE(), state)
info = spec.info
a, out = DeviceArray(info).set_valid(state.data), DeviceArray(info)
lib.set_ghost_cells(info.ref, 1, spec.bc_c.c, a.ptr, None)
e0, e1 = C.c_void_p(), C.c_void_p()
lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
ms = C.c_float()
best = 1e9
for _ in range(4):
    for _ in range(3):
        lib.laplace(info.ref, a.ptr, out.ptr, _abi.OUT_FULL, None)
    lib.stream_synchronize(None)
    lib.event_record(e0, None)
    for _ in range(40):
        lib.laplace(info.ref, a.ptr, out.ptr, _abi.OUT_FULL, None)
    lib.event_record(e1, None)
    lib.stream_synchronize(None)
    lib.event_elapsed_ms(e0, e1, C.byref(ms))
    best = min(best, ms.value / 40)
cells = n**3
tag = os.path.basename(os.environ.get("PDEHIP_LIB", "default"))
print(f"LAP {tag:>22s} n={n} {dtype}: {best:.4f} ms  {cells * 2 * dtype.itemsize / best / 1e6:.0f} GB/s  {cells * 2 * dtype.itemsize / best / 1e6 / 8000:.3f} of 8 TB/s")
