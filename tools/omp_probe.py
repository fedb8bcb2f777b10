import ctypes as C, os, sys, time
sys.path[:0] = ['.', 'py-pde_amd']
import numpy as np
from oracle import pde_oracle as O
from pde_hip import _abi
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try: print("cpu.max", open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e: print("no cpu.max", e)
gomp = C.CDLL("libgomp.so.1")
n = 256
g = _abi.make_grid((n,)*3, (1.0,)*3, np.float64)
faces = _abi.FaceArray()
for ax in range(3):
    for side, idx in ((0, n-1), (1, 0)):
        f = faces[2*ax+side]; f.kind, f.flags, f.index1, f.const_v, f.factor1 = _abi.BC_ORDER1, 0, idx, 0.0, 1.0
rhs = O.make_rhs(_abi.RHS_DIFFUSION, 1.0, faces)
a = O.valid_to_full((n,)*3, np.random.default_rng(0).random((n,)*3)); b = np.zeros_like(a)
res = C.c_void_p(); lib = O.lib()
for t in (4, 8, 16, 32, 64, 128, 256):
    if t > os.cpu_count(): break
    gomp.omp_set_num_threads(t)
    lib.oracle_euler_run(C.byref(g), C.byref(rhs), a.ctypes.data, b.ctypes.data, 0.1, 2, C.byref(res))
    t0 = time.perf_counter()
    lib.oracle_euler_run(C.byref(g), C.byref(rhs), a.ctypes.data, b.ctypes.data, 0.1, 6, C.byref(res))
    el = time.perf_counter() - t0
    print(f"threads {t:4d}: {n**3*6/el/1e6:9.1f} Mcells/s")
