import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "py-pde_amd"))
import pde_hip
from pde_hip.device import DeviceArray
n = 512
b = pde_hip.get_backend("hip"); lib = b._lib
grid = pde_hip.UnitGrid([n]*3, periodic=True)
state = pde_hip.ScalarField(grid, np.random.default_rng(0).uniform(-0.1, 0.1, grid.shape))
eq = pde_hip.PDE({"c": "c - c**3 + laplace(c)"})
erhs = b.make_expression_rhs(eq, state)
info = erhs.info
from pde_hip.device import DeviceBuffer
skew = int(os.environ.get("SKEW", "0"))
def alloc(j):
    if not skew:
        return DeviceArray(info)
    probe = DeviceArray(info)
    buf = DeviceBuffer(probe.nbytes + 16 * skew)
    return DeviceArray(info, buffer=buf, ptr=buf.ptr + j * skew)
y = alloc(0).set_valid(state.data)
k1, k2, k3, k4, tmp = [alloc(1 + j) for j in range(5)]
print("ptrs", [hex(a.ptr) for a in (y, k1, k2, k3, k4, tmp)])
ev = [C.c_void_p() for _ in range(6)]
for e in ev: lib.event_create(C.byref(e))
ms = C.c_float()
dt = 1e-3
def step(timed):
    lib.event_record(ev[0], b.stream)
    erhs.apply_stage(y, k1, dt, 0.0, 0, y, [], [], 0.5, tmp); lib.event_record(ev[1], b.stream)
    erhs.apply_stage(tmp, k2, dt, 0.0, 0, y, [], [], 0.5, k4); lib.event_record(ev[2], b.stream)
    erhs.apply_stage(k4, k3, dt, 0.0, 0, y, [], [], 1.0, tmp); lib.event_record(ev[3], b.stream)
    erhs.apply_stage(tmp, k4, dt, 0.0, 1, y, [k1, k2, k3], [], 0.0, y); lib.event_record(ev[4], b.stream)
    b.synchronize()
    if timed:
        out = []
        for i in range(4):
            lib.event_elapsed_ms(ev[i], ev[i+1], C.byref(ms)); out.append(ms.value)
        print("stage ms:", " ".join(f"{v:.3f}" for v in out), "sum", f"{sum(out):.3f}", flush=True)
step(False); step(False)
for _ in range(3): step(True)
# back-to-back without events in between
lib.event_record(ev[0], b.stream)
for _ in range(20):
    erhs.apply_stage(y, k1, dt, 0.0, 0, y, [], [], 0.5, tmp)
    erhs.apply_stage(tmp, k2, dt, 0.0, 0, y, [], [], 0.5, k4)
    erhs.apply_stage(k4, k3, dt, 0.0, 0, y, [], [], 1.0, tmp)
    erhs.apply_stage(tmp, k4, dt, 0.0, 1, y, [k1, k2, k3], [], 0.0, y)
lib.event_record(ev[1], b.stream); b.synchronize()
lib.event_elapsed_ms(ev[0], ev[1], C.byref(ms)); print("20 steps back to back:", ms.value / 20, "ms/step")
for name, fn in [("S1 only", lambda: erhs.apply_stage(y, k1, dt, 0.0, 0, y, [], [], 0.5, tmp)), ("S4 only", lambda: erhs.apply_stage(tmp, k4, dt, 0.0, 1, y, [k1, k2, k3], [], 0.0, k2))]:
    lib.event_record(ev[0], b.stream)
    for _ in range(20): fn()
    lib.event_record(ev[1], b.stream); b.synchronize()
    lib.event_elapsed_ms(ev[0], ev[1], C.byref(ms)); print(name, ms.value / 20, "ms")

out = DeviceArray(info)
for name, fn in [("plain jit_apply (scaled)", lambda: erhs.apply(y, out, "scaled", dt, 0.0))]:
    fn(); b.synchronize()
    lib.event_record(ev[0], b.stream)
    for _ in range(20): fn()
    lib.event_record(ev[1], b.stream); b.synchronize()
    lib.event_elapsed_ms(ev[0], ev[1], C.byref(ms)); print(name, ms.value / 20, "ms")
