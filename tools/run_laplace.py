"""Operator workload for the profiler: 60 applications of the 3-D fp64 Laplacian (pdehip_laplace, register-pipelined one-level
kernel) and of gradient / divergence on a resident 512^3 field.  usage: python tools/run_laplace.py [n=512]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import numpy as np

import pde_hip
from pde_hip import _abi
from pde_hip.device import DeviceArray

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
b = pde_hip.get_backend("hip")
lib = b._lib
grid = pde_hip.UnitGrid([n] * 3, periodic=True)
state = pde_hip.ScalarField(grid, np.random.default_rng(0).random((n,) * 3))
spec = b.make_rhs_spec(pde_hip.DiffusionPDE(), state)
info = spec.info
a, out = DeviceArray(info).set_valid(state.data), DeviceArray(info)
lib.set_ghost_cells(info.ref, 1, spec.bc_c.c, a.ptr, None)
for _ in range(60):
    lib.laplace(info.ref, a.ptr, out.ptr, _abi.OUT_FULL, None)
lib.stream_synchronize(None)
print("done", float(out.get_valid()[0, 0, 0]))
