"""Randomised parity check of the two-level kernel against the oracle (run on a GPU box; not part of the test-suite).

usage: python tools/fuzz_two_level.py [cases] [seed]
Random 2-D / 3-D shapes (ANY row length and row count, few or many planes), dtypes, periodic / local faces with random
coefficients (different ones for c and mu), all three fused modes, one RK4 step and one RKF45 attempt of both right-hand sides
(stage epilogues).  Bit-exact or it prints the case.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "py-pde_amd"), os.path.join(ROOT, "tests")]
import pde_hip  # noqa: E402
from helpers import host_faces, interior, oracle_grid, to_full  # noqa: E402
from oracle import pde_oracle as O  # noqa: E402
from pde_hip import _abi  # noqa: E402
from pde_hip.backend import convert_bcs  # noqa: E402
from pde_hip.device import DeviceArray, GridInfo  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib = pde_hip.get_backend("hip")._lib


def rand_face(r):
    kind = r.integers(3)
    if kind == 0:
        return {"value": float(r.uniform(-1, 1))}
    if kind == 1:
        return {"derivative": float(r.uniform(-1, 1))}
    return {"type": "mixed", "value": float(r.uniform(-2, 2)), "const": float(r.uniform(-1, 1))}


bad = covered = 0
for case in range(cases):
    ndim = int(rng.choice([2, 3]))
    dtype = np.dtype(rng.choice(["float64", "float32"]))
    vec = 16 // dtype.itemsize
    # any row length (rows that end inside a vector: the last chunk is moved back over its neighbour) and any row count
    n2 = int(rng.integers(4, 640)) if rng.random() < 0.7 else int(rng.choice([128, 129, 255, 256, 257, 384, 512, 513]))
    shape = ([int(rng.integers(4, 40)), int(rng.integers(4, 42)), n2] if ndim == 3 else [int(rng.integers(4, 60)), n2])
    periodic = [bool(rng.integers(2)) for _ in range(ndim)]
    grid = pde_hip.CartesianGrid([[0, n * float(rng.uniform(0.5, 1.5))] for n in shape], shape, periodic=periodic)
    bcs = []
    for _ in range(2):
        bc = {}
        for i, ax in enumerate(grid.axes):
            if periodic[i]:
                bc[ax] = "periodic"
            else:
                bc[ax + "-"], bc[ax + "+"] = rand_face(rng), rand_face(rng)
        bcs.append(grid.get_boundary_conditions(bc))
    data = rng.uniform(-0.5, 0.5, shape).astype(dtype)
    info = GridInfo(grid.shape, grid.discretization, dtype)
    fc, fm = convert_bcs(bcs[0]), convert_bcs(bcs[1])
    g = oracle_grid(grid, dtype)
    a, b = DeviceArray(info).set_valid(data), DeviceArray(info)
    done = C.c_int(0)
    D, dt, gamma = float(rng.uniform(0.1, 1)), float(rng.uniform(1e-4, 1e-2)), float(rng.uniform(0.2, 1.5))
    rd = O.make_rhs(_abi.RHS_DIFFUSION, D, host_faces(bcs[0]).c)
    scratch = np.zeros(grid._shape_full, dtype)
    rch = O.make_rhs(_abi.RHS_CAHN_HILLIARD, gamma, host_faces(bcs[0]).c, host_faces(bcs[1]).c, scratch)
    checks = []
    lib.diffusion_euler2(info.ref, fc.c, a.ptr, b.ptr, D, dt, C.byref(done), None)
    if done.value:
        checks.append(("diffusion x2", b.get_valid(), interior(grid, O.euler_run(g, rd, to_full(grid, data), dt, 2))))
    lib.cahn_hilliard_fused(info.ref, fc.c, fm.c, a.ptr, b.ptr, gamma, dt, 1, C.byref(done), None)
    if done.value:
        checks.append(("CH euler", b.get_valid(), interior(grid, O.euler_run(g, rch, to_full(grid, data), dt, 1))))
    lib.cahn_hilliard_fused(info.ref, fc.c, fm.c, a.ptr, b.ptr, gamma, dt, 0, C.byref(done), None)
    if done.value:
        checks.append(("CH scaled", b.get_valid(), interior(grid, O.rhs_scaled(g, rch, to_full(grid, data), dt))))
    # Runge-Kutta sweeps (stage epilogues of both right-hand sides; RK4 writes the new state in place): one RK4 step, one RKF45 attempt
    from pde_hip.device import DeviceScalar, ptr_array  # noqa: E402

    for kind, orhs in (("diffusion", rd), ("cahn_hilliard", rch)):
        spec = _abi.RHS()
        spec.kind = _abi.RHS_DIFFUSION if kind == "diffusion" else _abi.RHS_CAHN_HILLIARD
        spec.param = D if kind == "diffusion" else gamma
        fc.copy_into(spec.bc_c)
        mu = DeviceArray(info)
        if kind != "diffusion":
            fm.copy_into(spec.bc_mu)
            spec.scratch_mu = mu.ptr
        y, ynew, err = DeviceArray(info).set_valid(data), DeviceArray(info), DeviceScalar()
        work = [DeviceArray(info) for _ in range(7)]
        lib.rkf45_attempt(info.ref, C.byref(spec), y.ptr, ynew.ptr, ptr_array(work), dt, err.ptr, None)
        want_new, want_err = O.rkf45_attempt(g, orhs, to_full(grid, data), dt)
        checks.append((f"{kind} rkf45", ynew.get_valid(), interior(grid, want_new)))
        if err.value() != want_err and not (np.isnan(err.value()) and np.isnan(want_err)):
            bad += 1
            print(f"MISMATCH case {case}: {kind} rkf45 error norm {err.value()} != {want_err} shape={shape} {dtype}", flush=True)
        lib.rk4_step(info.ref, C.byref(spec), y.ptr, ptr_array(work[:5]), dt, None)
        yo = to_full(grid, data)
        O.rk4_step(g, orhs, yo, dt)
        checks.append((f"{kind} rk4", y.get_valid(), interior(grid, yo)))
    covered += bool(checks)
    for name, got, want in checks:
        if not np.array_equal(got, want, equal_nan=True):
            bad += 1
            print(f"MISMATCH case {case}: {name} shape={shape} {dtype} periodic={periodic} max|d|={np.abs(got - want).max():.3e}", flush=True)
print(f"{cases} cases, {covered} covered by the two-level kernel, {bad} mismatches")
sys.exit(1 if bad else 0)
