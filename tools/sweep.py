"""Tuning aid: time the product's fused Euler kernel / whole Euler step for several tile shapes.

usage: python tools/sweep.py [n]        (spawns one process per PDEHIP_TUNE setting)
"""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CONFIGS = ["default", "4,4,1,1,256", "4,4,1,1,512", "4,4,1,1,1024", "2,4,1,1,512", "2,4,1,1,1024", "2,4,1,1,2048",
           "2,2,4,2,256", "2,2,4,2,512", "2,4,4,1,256", "2,4,4,1,512", "2,1,4,1,512"]


def worker(n: int) -> None:
    sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
    import numpy as np

    import pde_hip
    from pde_hip.device import DeviceArray

    b = pde_hip.get_backend("hip")
    lib = b._lib
    grid = pde_hip.UnitGrid([n, n, n], periodic=True)
    dtype = np.dtype(os.environ.get("SWEEP_DTYPE", "float64"))
    state = pde_hip.ScalarField(grid, np.random.default_rng(0).random((n, n, n)), dtype=dtype)
    spec = b.make_rhs_spec(pde_hip.DiffusionPDE(), state)
    info = spec.info
    a, bb = DeviceArray(info).set_valid(state.data), DeviceArray(info)
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        lib.event_create(C.byref(e))
    res = C.c_void_p()
    ms = C.c_float()

    def timed(fn, reps):
        fn(3)
        lib.stream_synchronize(None)
        lib.event_record(ev[0], None)
        fn(reps)
        lib.event_record(ev[1], None)
        lib.stream_synchronize(None)
        lib.event_elapsed_ms(ev[0], ev[1], C.byref(ms))
        return ms.value / reps

    def kern(k):
        for _ in range(k):
            lib.laplace_euler(info.ref, a.ptr, a.ptr, bb.ptr, 1.0, 0.1, None)

    def plain(k):
        for _ in range(k):
            lib.laplace(info.ref, a.ptr, bb.ptr, 1, None)

    def step(k):
        lib.euler_run(info.ref, spec.ref, a.ptr, bb.ptr, 0.1, k, C.byref(res), None)

    tk, tp, ts = timed(kern, 50), timed(plain, 50), timed(step, 100)
    cells = n**3
    print(f"{os.environ.get('PDEHIP_TUNE', 'default'):>16}: euler kernel {tk:.4f} ms ({cells*2*dtype.itemsize/tk/1e6/8000*100:.1f}%)  "
          f"laplace {tp:.4f} ms ({cells*2*dtype.itemsize/tp/1e6/8000*100:.1f}%)  euler step {ts:.4f} ms ({cells/ts/1e6:.1f} Gcells/s, {cells*2*dtype.itemsize/ts/1e6/8000*100:.1f}%)", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "worker":
        worker(int(sys.argv[1]))
    else:
        n = sys.argv[1] if len(sys.argv) > 1 else "512"
        for cfg in CONFIGS:
            env = dict(os.environ)
            if cfg != "default":
                env["PDEHIP_TUNE"] = cfg
            subprocess.run([sys.executable, __file__, n, "worker"], env=env, check=False)
