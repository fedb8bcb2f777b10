"""Follow-up of tools/probe_modes.py: both arrays of the ping-pong pair carved out of ONE allocation at a chosen distance.
Is the timing mode a function of that distance (then it can be chosen), or of the physical pages (then it cannot)?"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import numpy as np

import pde_hip
from pde_hip.device import DeviceArray, DeviceBuffer

n = 512
b = pde_hip.get_backend("hip")
lib = b._lib
grid = pde_hip.UnitGrid([n] * 3, periodic=True)
state = pde_hip.ScalarField(grid, np.random.default_rng(0).random((n,) * 3))
spec = b.make_rhs_spec(pde_hip.DiffusionPDE(), state)
info = spec.info
nbytes = DeviceArray(info).nbytes
ev = [C.c_void_p() for _ in range(2)]
for e in ev:
    lib.event_create(C.byref(e))
MiB = 1 << 20
size = (nbytes + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)
skews = [0, 4096, 65536, 256 * 1024, MiB, 2 * MiB + 4096, 7 * MiB, 16 * MiB, 33 * MiB + 8192, 64 * MiB]
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    big = DeviceBuffer(2 * size + 128 * MiB)
    lib.memset(big.ptr, 0, big.nbytes, None)
    line = []
    for skew in skews:
        a = DeviceArray(info, buffer=big, ptr=big.ptr)
        bb = DeviceArray(info, buffer=big, ptr=big.ptr + size + skew)
        a.set_valid(state.data)
        res = C.c_void_p()
        lib.euler_run(info.ref, spec.ref, a.ptr, bb.ptr, 0.1, 20, C.byref(res), None)
        lib.stream_synchronize(None)
        lib.event_record(ev[0], None)
        lib.euler_run(info.ref, spec.ref, a.ptr, bb.ptr, 0.1, 100, C.byref(res), None)
        lib.event_record(ev[1], None)
        lib.stream_synchronize(None)
        ms = C.c_float()
        lib.event_elapsed_ms(ev[0], ev[1], C.byref(ms))
        line.append(f"{skew / MiB:.3f}:{ms.value / 50:.4f}")
    print(f"allocation {trial} at {big.ptr:#x}: skew MiB : ms per launch   " + "  ".join(line), flush=True)
    keep = big if trial % 2 == 0 else None     # (hold every other one, so that the next allocation gets other memory)
