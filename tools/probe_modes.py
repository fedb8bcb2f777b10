"""Do the two timing modes of the two-step sweep depend on WHERE the two arrays of the ping-pong pair live?
One process, the pair allocated several times (kept alive, so that every round gets other memory), the sweep timed each time."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import numpy as np

import pde_hip
from pde_hip.device import DeviceArray

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
b = pde_hip.get_backend("hip")
lib = b._lib
grid = pde_hip.UnitGrid([n] * 3, periodic=True)
state = pde_hip.ScalarField(grid, np.random.default_rng(0).random((n,) * 3))
spec = b.make_rhs_spec(pde_hip.DiffusionPDE(), state)
info = spec.info
ev = [C.c_void_p() for _ in range(2)]
for e in ev:
    lib.event_create(C.byref(e))
keep = []
for r in range(rounds):
    a, bb = DeviceArray(info).set_valid(state.data), DeviceArray(info)
    keep.append((a, bb))
    res = C.c_void_p()
    times = []
    for rep in range(3):
        lib.euler_run(info.ref, spec.ref, a.ptr, bb.ptr, 0.1, 20, C.byref(res), None)
        lib.stream_synchronize(None)
        lib.event_record(ev[0], None)
        lib.euler_run(info.ref, spec.ref, a.ptr, bb.ptr, 0.1, 200, C.byref(res), None)
        lib.event_record(ev[1], None)
        lib.stream_synchronize(None)
        ms = C.c_float()
        lib.event_elapsed_ms(ev[0], ev[1], C.byref(ms))
        times.append(ms.value / 100)
    print(f"pair {r}: a = {a.ptr:#x}  b = {bb.ptr:#x}  b - a = {(bb.ptr - a.ptr) / 2**20:.3f} MiB   ms per two-step launch: " + "  ".join(f"{t:.4f}" for t in times), flush=True)
    if len(keep) > 3:
        keep.pop(0)
