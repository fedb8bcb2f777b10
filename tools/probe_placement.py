"""Does the relative placement of the two state buffers matter for the two-step sweep?  (same-process probe: one big allocation,
the second buffer at varying byte offsets behind the first).  usage: python tools/probe_placement.py [n]"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import numpy as np

import pde_hip
from pde_hip.device import DeviceArray, DeviceBuffer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
b = pde_hip.get_backend("hip")
lib = b._lib
grid = pde_hip.UnitGrid((n,) * 3, periodic=True)
eq = pde_hip.DiffusionPDE()
state = pde_hip.ScalarField(grid, np.random.default_rng(0).random((n,) * 3))
spec = b.make_rhs_spec(eq, state)
info = spec.info
probe = DeviceArray(info)
nbytes = probe.nbytes
size2m = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
big = DeviceBuffer(2 * size2m + (160 << 20))
ev = [C.c_void_p() for _ in range(2)]
for e in ev:
    lib.event_create(C.byref(e))
print(f"# {n}^3 fp64, array {nbytes / 2**20:.1f} MiB, base 0x{big.ptr:x}")
print("| offset of buffer B behind the 2 MiB-rounded end of A | Euler ms/step |")
print("|---:|---:|")
for delta in [0, 256, 1024, 4096, 8192, 16384, 65536, 1 << 18, 1 << 20, 2 << 20, 3 << 20, 8 << 20, 32 << 20, (32 << 20) + 4096, 128 << 20, 0]:
    a = DeviceArray(info, buffer=big, ptr=big.ptr).set_valid(state.data)
    bb = DeviceArray(info, buffer=big, ptr=big.ptr + size2m + delta)
    res = C.c_void_p()
    steps = 40
    best = 1e9
    for rep in range(4):
        lib.stream_synchronize(None)
        lib.event_record(ev[0], None)
        lib.euler_run(info.ref, spec.ref, a.ptr, bb.ptr, 0.1, steps, C.byref(res), None)
        lib.event_record(ev[1], None)
        lib.stream_synchronize(None)
        ms = C.c_float()
        lib.event_elapsed_ms(ev[0], ev[1], C.byref(ms))
        if rep:
            best = min(best, ms.value / steps)
    print(f"| {delta} | {best:.4f} |", flush=True)
