"""Host <-> device transfer rates of a field's valid data (pageable numpy memory, contiguous array and `field.data`-like window).
usage: python tools/time_transfers.py [n0xn1xn2 ...]
"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import numpy as np

import pde_hip
from pde_hip.device import DeviceArray, DeviceBuffer, GridInfo

b = pde_hip.get_backend("hip")
lib = b._lib
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(512, 512, 512), (256, 256, 256), (1024, 1024)]
print("| grid (fp64) | MB | what | ms | GB/s |")
print("|---|---:|---|---:|---:|")


def best(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


for shape in shapes:
    info = GridInfo(shape, (1.0,) * len(shape), np.dtype(np.float64))
    rng = np.random.default_rng(0)
    full = rng.random(tuple(s + 2 for s in shape))
    view = full[tuple(slice(1, -1) for _ in shape)]
    cont = np.ascontiguousarray(view)
    out = np.empty_like(cont)
    out[...] = 0   # pages touched
    dev = DeviceArray(info)
    mb = cont.nbytes / 1e6
    buf = DeviceBuffer(cont.nbytes)
    rows = [
        ("memcpy_h2d contiguous", lambda: lib.memcpy_h2d(buf.ptr, cont.ctypes.data, cont.nbytes, None)),
        ("memcpy_d2h contiguous", lambda: lib.memcpy_d2h(out.ctypes.data, buf.ptr, cont.nbytes, None)),
        ("set_valid(contiguous)  [+ layout kernel]", lambda: dev.set_valid(cont)),
        ("set_valid(field.data window)", lambda: dev.set_valid(view)),
        ("get_valid(out=contiguous)", lambda: dev.get_valid(out=out)),
        ("get_valid(out=field.data window)", lambda: dev.get_valid(out=view)),
        ("get_valid() into fresh pages", lambda: dev.get_valid()),
        ("host only: np.ascontiguousarray(window)", lambda: np.ascontiguousarray(view)),
        ("host only: window[...] = contiguous", lambda: view.__setitem__(Ellipsis, cont)),
    ]
    for name, fn in rows:
        t = best(fn)
        print(f"| {'x'.join(map(str, shape))} | {mb:.0f} | {name} | {t * 1e3:.1f} | {mb / 1e3 / t:.1f} |", flush=True)
    buf.free()
