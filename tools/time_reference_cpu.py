"""Time the REFERENCE itself on the headline workload (SURVEY.md §8d "CPU side-by-side").

numba is not installable here, so the reference paths that run are (i) its eager torch-CPU backend and (ii) its numpy
backend with scipy operators; the CPU oracle (oracle/pde_oracle.c, the C port of the numba formulas) is timed on the
same cores for an apples-to-apples ratio.  Runs in the build container (reference at /root/reference) or, once per round, on
the GPU box's host cores with the reference shipped as untracked scratch (`PDEHIP_REFERENCE=<dir>`, tools/gpu_r3_dropin.sh);
writes the JSON that bench.py attaches to its line as `cpu_baseline_reference` (labelled with where it was measured).

usage: [PDEHIP_REFERENCE=dir] python tools/time_reference_cpu.py [n=512] [steps=6] [output.json]
"""
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
REF = os.environ.get("PDEHIP_REFERENCE") or "/root/reference"
sys.path.append(REF)
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cores = len(os.sched_getaffinity(0))
try:
    quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
    if quota != "max":
        cores = max(1, min(cores, int(int(quota) / int(period))))
except (OSError, ValueError):
    pass

import pde  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(cores)
pde.config["backend.torch.compile"] = False
grid = pde.UnitGrid([n] * 3, periodic=True)
state = pde.ScalarField.random_uniform(grid, rng=np.random.default_rng(0))
eq = pde.DiffusionPDE()
out = {"workload": f"DiffusionPDE(D=1) on UnitGrid([{n}]*3, periodic=True) fp64, explicit Euler dt=0.1", "cores": cores,
       "where": os.environ.get("PDEHIP_WHERE") or "build container (the reference cannot travel to the GPU box)", "unit": "Mcells/s"}

# (i) reference, eager torch-CPU backend: time the stepper alone (Controller profiler), warm
eq.solve(state, t_range=0.1, dt=0.1, backend="torch", solver="euler", tracker=None)
t0 = time.perf_counter()
_, info = eq.solve(state, t_range=0.1 * steps, dt=0.1, backend="torch", solver="euler", tracker=None, ret_info=True)
wall = time.perf_counter() - t0
out["reference_torch_cpu_eager"] = {"value": round(n**3 * steps / wall / 1e6, 1), "steps": steps, "seconds": round(wall, 2), "kind": "reference",
                                    "note": "py-pde torch backend, device cpu, compile=False; wall of eq.solve incl. its host copies"}
print(out["reference_torch_cpu_eager"], flush=True)

# (ii) reference, numpy backend with scipy operators (the only reference path that also runs RK4 / RKF45 here)
pde.config["default_backend"] = "scipy"
t0 = time.perf_counter()
eq.solve(state, t_range=0.1 * 2, dt=0.1, backend="numpy", solver="euler", tracker=None)
wall = time.perf_counter() - t0
out["reference_numpy_scipy"] = {"value": round(n**3 * 2 / wall / 1e6, 1), "steps": 2, "seconds": round(wall, 2), "kind": "reference"}
print(out["reference_numpy_scipy"], flush=True)

# (iii) the C port on the same cores
from oracle import pde_oracle as O  # noqa: E402
from pde_hip import _abi  # noqa: E402

C.CDLL("libgomp.so.1").omp_set_num_threads(cores)
g = _abi.make_grid((n,) * 3, (1.0,) * 3, np.float64)
faces = _abi.FaceArray()
for ax in range(3):
    for side, idx in ((0, n - 1), (1, 0)):
        f = faces[2 * ax + side]
        f.kind, f.flags, f.index1, f.const_v, f.factor1 = _abi.BC_ORDER1, 0, idx, 0.0, 1.0
rhs = O.make_rhs(_abi.RHS_DIFFUSION, 1.0, faces)
a = O.valid_to_full((n,) * 3, state.data)
b = np.zeros_like(a)
res = C.c_void_p()
lib = O.lib()
lib.oracle_euler_run(C.byref(g), C.byref(rhs), a.ctypes.data, b.ctypes.data, 0.1, 2, C.byref(res))
t0 = time.perf_counter()
k = 20
lib.oracle_euler_run(C.byref(g), C.byref(rhs), a.ctypes.data, b.ctypes.data, 0.1, k, C.byref(res))
wall = time.perf_counter() - t0
out["oracle_port_same_cores"] = {"value": round(n**3 * k / wall / 1e6, 1), "steps": k, "seconds": round(wall, 2), "kind": "port"}
print(out["oracle_port_same_cores"], flush=True)
target = Path(sys.argv[3]) if len(sys.argv) > 3 else ROOT / "profiles" / "reference_cpu.json"
target.write_text(json.dumps(out, indent=1) + "\n")
