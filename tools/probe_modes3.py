"""Follow-up of tools/probe_modes.py: ONE array `a` kept, the partner `b` taken from several allocations - does the timing mode belong to
the pair?  Then the reverse: one `b`, several `a`."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import numpy as np

import pde_hip
from pde_hip.device import DeviceArray

n = 512
b = pde_hip.get_backend("hip")
lib = b._lib
grid = pde_hip.UnitGrid([n] * 3, periodic=True)
state = pde_hip.ScalarField(grid, np.random.default_rng(0).random((n,) * 3))
spec = b.make_rhs_spec(pde_hip.DiffusionPDE(), state)
info = spec.info
ev = [C.c_void_p() for _ in range(2)]
for e in ev:
    lib.event_create(C.byref(e))


def timed(a, bb, steps=60):
    a.set_valid(state.data)
    res = C.c_void_p()
    lib.euler_run(info.ref, spec.ref, a.ptr, bb.ptr, 0.1, 10, C.byref(res), None)
    lib.stream_synchronize(None)
    lib.event_record(ev[0], None)
    lib.euler_run(info.ref, spec.ref, a.ptr, bb.ptr, 0.1, steps, C.byref(res), None)
    lib.event_record(ev[1], None)
    lib.stream_synchronize(None)
    ms = C.c_float()
    lib.event_elapsed_ms(ev[0], ev[1], C.byref(ms))
    return ms.value / (steps // 2)


arrays = [DeviceArray(info) for _ in range(6)]
print("addresses (GiB offsets from the lowest):", [f"{(x.ptr - min(y.ptr for y in arrays)) / 2**30:.2f}" for x in arrays])
print("ms per two-step launch, row = array read first (a), column = partner (b):")
for i, a in enumerate(arrays):
    print(f"a{i}: " + "  ".join("  --  " if i == j else f"{timed(a, bb):.4f}" for j, bb in enumerate(arrays)), flush=True)
