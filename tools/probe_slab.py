"""Overhead probe for the slab stepper on ONE GPU: a slab-sized periodic grid whose axis-0 halo is
sent to self through RCCL (same code path as N ranks).  Reports ms/step, host enqueue time per step
and the single-kernel reference (pdehip_euler_run) for the same grid.
usage: python tools/probe_slab.py 64,512,512 [steps]
"""
import ctypes as C
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import numpy as np

import pde_hip
from pde_hip.device import DeviceArray
from pde_hip.distributed import SlabStepper

shape = tuple(int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "64,512,512").split(","))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
grid = pde_hip.UnitGrid(shape, periodic=True)
eq = pde_hip.DiffusionPDE(1.0)
cells = int(np.prod(shape))
for force in (True, False):
    st = SlabStepper(eq, grid, force_exchange=force)
    cur, nxt = st.buf("state_a"), st.buf("state_b")
    st.set_local(cur, np.random.default_rng(0).random(shape))
    cur = st.euler_steps(cur, nxt, 0.1, 20)
    nxt = st.buf("state_b") if cur is st.buf("state_a") else st.buf("state_a")
    st.synchronize()
    t0 = time.perf_counter()
    cur = st.euler_steps(cur, nxt, 0.1, steps)
    t_enq = time.perf_counter() - t0
    st.synchronize()
    t_all = time.perf_counter() - t0
    st.close()
    print(f"{shape} slab stepper exchange={force}: {t_all/steps*1e3:.4f} ms/step ({cells*steps/t_all/1e9:.1f} Gcells/s), host enqueue {t_enq/steps*1e6:.1f} us/step", flush=True)
b = pde_hip.get_backend("hip")
state = pde_hip.ScalarField(grid, np.random.default_rng(0).random(shape))
spec = b.make_rhs_spec(eq, state)
a, bb = DeviceArray(spec.info).set_valid(state.data), DeviceArray(spec.info)
res = C.c_void_p()
b._lib.euler_run(spec.info.ref, spec.ref, a.ptr, bb.ptr, 0.1, 20, C.byref(res), None)
b._lib.stream_synchronize(None)
t0 = time.perf_counter()
b._lib.euler_run(spec.info.ref, spec.ref, a.ptr, bb.ptr, 0.1, steps, C.byref(res), None)
t_enq = time.perf_counter() - t0
b._lib.stream_synchronize(None)
t_all = time.perf_counter() - t0
print(f"{shape} pdehip_euler_run (single kernel/step): {t_all/steps*1e3:.4f} ms/step ({cells*steps/t_all/1e9:.1f} Gcells/s), host enqueue {t_enq/steps*1e6:.1f} us/step")
