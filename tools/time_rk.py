"""Time one RK4 step and one RKF45 attempt of the built-in right-hand sides through the C ABI.

usage: python tools/time_rk.py [size] [diffusion|cahn_hilliard] [periodic|neumann]
Set PDEHIP_RK_FUSE=0 to time the unfused sequence (separate lincomb / combine kernels) for comparison.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "py-pde_amd"))
import pde_hip  # noqa: E402
from pde_hip.device import DeviceArray, DeviceBuffer, DeviceScalar, ptr_array  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = sys.argv[2] if len(sys.argv) > 2 else "diffusion"
bcname = sys.argv[3] if len(sys.argv) > 3 else "periodic"
backend = pde_hip.get_backend("hip")
lib = backend._lib
grid = pde_hip.UnitGrid([n, n, n], periodic=bcname == "periodic")
bc = "auto_periodic_neumann"
eq = pde_hip.DiffusionPDE(0.5, bc=bc) if kind == "diffusion" else pde_hip.CahnHilliardPDE(1.0, bc_c=bc, bc_mu=bc)
state = pde_hip.ScalarField(grid, np.random.default_rng(0).uniform(-0.1, 0.1, grid.shape))
spec = backend.make_rhs_spec(eq, state)
info = spec.info
skew = int(os.environ.get("SKEW", "0"))   # bytes added to the start of the j-th array (DRAM channel / bank alignment probe)


def alloc(j):
    if not skew:
        return DeviceArray(info)
    probe = DeviceArray(info)
    buf = DeviceBuffer(probe.nbytes + 16 * skew)
    return DeviceArray(info, buffer=buf, ptr=buf.ptr + j * skew)


y, ynew, err = alloc(0).set_valid(state.data), alloc(1), DeviceScalar()
work = [alloc(2 + j) for j in range(7)]
wp = ptr_array(work)
e0, e1 = C.c_void_p(), C.c_void_p()
lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
ms = C.c_float()


def timeit(fn, name, reps=10):
    fn()
    lib.stream_synchronize(None)
    lib.event_record(e0, None)
    for _ in range(reps):
        fn()
    lib.event_record(e1, None)
    lib.stream_synchronize(None)
    lib.event_elapsed_ms(e0, e1, C.byref(ms))
    print(f"| {n}^3 {kind} {bcname} | {name} | fuse={os.environ.get('PDEHIP_RK_FUSE', '1')} skew={skew} | {ms.value / reps:.4f} ms |", flush=True)


dt = 1e-6
reps = int(os.environ.get("REPS", "10"))
timeit(lambda: lib.rk4_step(info.ref, spec.ref, y.ptr, wp, dt, None), "rk4_step")
timeit(lambda: lib.rkf45_attempt(info.ref, spec.ref, y.ptr, ynew.ptr, wp, dt, err.ptr, None), "rkf45_attempt")

# Adams-Bashforth step: one sweep (pdehip_ab2_step) vs rate + combine kernels
rates = [work[0], work[1]]
nxt = work[2]
fused = C.c_int(0)


def ab_fused():
    lib.ab2_step(info.ref, spec.ref, y.ptr, nxt.ptr, rates[0].ptr, rates[1].ptr, dt, C.byref(fused), None)


def ab_separate():
    lib.rhs_scaled(info.ref, spec.ref, y.ptr, rates[0].ptr, 1.0, None)
    lib.ab2_combine(info.ref, 1, y.ptr, rates[0].ptr, rates[1].ptr, dt, None)


ab_fused()
timeit(ab_fused if fused.value else ab_separate, "adams-bashforth step, one sweep" if fused.value else "adams-bashforth step (sweep refused)")
timeit(ab_separate, "adams-bashforth step, rate + combine kernels")
