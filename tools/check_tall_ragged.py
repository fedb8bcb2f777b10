"""Parity of the tall tile with the ragged-row code (build-time experiment PDEHIP_TALL_RAGGED, tools/build_variant.sh tallr) against the oracle:
two Euler steps per sweep on grids whose rows / row counts are not multiples of the tile, every combination of periodic / local faces.
usage: PDEHIP_LIB=tools/variants/libpdehip_tallr.so PDEHIP_EULER2=8 python tools/check_tall_ragged.py"""
import ctypes as C
import itertools
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd"), str(ROOT / "tests")]
import numpy as np
from helpers import host_faces, interior, oracle_grid, to_full

import pde_hip
from oracle import pde_oracle as O
from pde_hip import _abi
from pde_hip.backend import convert_bcs
from pde_hip.device import DeviceArray, GridInfo

LOCAL = [({"value": 0.5}, {"derivative": 0.25}), ({"type": "mixed", "value": 1.5, "const": 0.2}, {"value": -0.3}),
         ({"derivative": -1.0}, {"type": "mixed", "value": -0.5, "const": 1.0})]
backend = pde_hip.get_backend("hip")
bad = 0
for shape in [(9, 64, 128), (7, 67, 131), (12, 65, 257), (5, 100, 513), (6, 72, 200), (10, 69, 136), (4, 64, 129), (5, 71, 384)]:
    for periodic in itertools.product([True, False], repeat=3):
        grid = pde_hip.CartesianGrid([[0, n * (0.7 + 0.1 * i)] for i, n in enumerate(shape)], shape, periodic=periodic)
        bc = {}
        for i, a in enumerate(grid.axes):
            if periodic[i]:
                bc[a] = "periodic"
            else:
                bc[a + "-"], bc[a + "+"] = LOCAL[i]
        bcs = grid.get_boundary_conditions(bc)
        data = np.random.default_rng(3).uniform(-0.5, 0.5, shape)
        info = GridInfo(grid.shape, grid.discretization, data.dtype)
        a, b = DeviceArray(info).set_valid(data), DeviceArray(info)
        done = C.c_int(0)
        backend._lib.diffusion_euler2(info.ref, convert_bcs(bcs).c, a.ptr, b.ptr, 0.8, 0.05, C.byref(done), None)
        g = oracle_grid(grid, data.dtype)
        rhs = O.make_rhs(_abi.RHS_DIFFUSION, 0.8, host_faces(bcs).c)
        expect = interior(grid, O.euler_run(g, rhs, to_full(grid, data), 0.05, 2))
        ok = bool(done.value) and np.array_equal(b.get_valid(), expect)
        bad += not ok
        if not ok:
            print("MISMATCH", shape, periodic, done.value, float(np.abs(b.get_valid() - expect).max()))
print("tall-ragged parity:", "ok" if not bad else f"{bad} mismatches")
