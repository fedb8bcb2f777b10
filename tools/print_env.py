"""Environment as the C runtime sees it AFTER the HIP runtime (and any preloaded tool library) initialised."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "py-pde_amd"))
before = dict(os.environ)
import pde_hip  # noqa: E402

b = pde_hip.get_backend("hip")
b.synchronize()
libc = C.CDLL(None)
env = C.POINTER(C.c_char_p).in_dll(libc, "environ")
i = 0
now = {}
while env[i]:
    k, _, v = env[i].decode(errors="replace").partition("=")
    now[k] = v
    i += 1
for k in sorted(now):
    if k not in before or before[k] != now[k]:
        print("CHANGED", k, "=", now[k][:200])
print("total", len(now), "vars; changed listed above")
