"""One process = one choice of the fp32 wave tile of the two-level kernel (``PDEHIP_F32_TILE="vec,ry,stage_vec,stage_ry"`` is read
once per process): Euler runs, the fused Cahn-Hilliard sweeps, an RK4 step and an RKF45 attempt of fp32 3-D grids through the C
ABI, bit for bit against the oracle.  Launched by tests/test_hip_euler2.py::test_fp32_tile_shapes."""
from __future__ import annotations

import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "py-pde_amd", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import pde_hip  # noqa: E402
from helpers import host_faces, interior, oracle_grid, to_full  # noqa: E402
from oracle import pde_oracle as O  # noqa: E402
from pde_hip import _abi  # noqa: E402
from pde_hip.device import DeviceArray, DeviceScalar, ptr_array  # noqa: E402


def main() -> int:
    backend = pde_hip.get_backend("hip")
    lib = backend._lib
    rng = np.random.default_rng(5)
    dtype = np.float32
    failures = []
    cases = [((12, 8, 256), [True, True, True]), ((9, 12, 200), [True, False, False]), ((10, 6, 264), [False, True, False]),
             ((7, 4, 64), [True, True, True])]
    for shape, periodic in cases:
        grid = pde_hip.CartesianGrid([[0, n * 0.9] for n in shape], shape, periodic=periodic)
        bc = "auto_periodic_neumann"
        bcs = grid.get_boundary_conditions(bc)
        data = rng.uniform(-0.5, 0.5, shape).astype(dtype)
        g = oracle_grid(grid, dtype)
        hf = host_faces(bcs)
        for kind in ("diffusion", "cahn_hilliard"):
            scratch = np.zeros(grid._shape_full, dtype)
            if kind == "diffusion":
                orhs, eq = O.make_rhs(_abi.RHS_DIFFUSION, 0.6, hf.c), pde_hip.DiffusionPDE(0.6, bc=bc)
            else:
                orhs, eq = O.make_rhs(_abi.RHS_CAHN_HILLIARD, 0.9, hf.c, hf.c, scratch), pde_hip.CahnHilliardPDE(0.9, bc_c=bc, bc_mu=bc)
            dt = 0.002
            spec = backend.make_rhs_spec(eq, pde_hip.ScalarField(grid, data, dtype=dtype))
            info = spec.info
            a, b = DeviceArray(info).set_valid(data), DeviceArray(info)
            res = C.c_void_p()
            lib.euler_run(info.ref, spec.ref, a.ptr, b.ptr, dt, 7, C.byref(res), None)
            got = (b if res.value == b.ptr else a).get_valid()
            if not np.array_equal(got, interior(grid, O.euler_run(g, orhs, to_full(grid, data), dt, 7))):
                failures.append(f"{shape} {kind}: euler_run")
            y = DeviceArray(info).set_valid(data)
            work = [DeviceArray(info) for _ in range(7)]
            lib.rk4_step(info.ref, spec.ref, y.ptr, ptr_array(work[:5]), dt, None)
            yo = to_full(grid, data)
            O.rk4_step(g, orhs, yo, dt)
            if not np.array_equal(y.get_valid(), interior(grid, yo)):
                failures.append(f"{shape} {kind}: rk4_step")
            y.set_valid(data)
            ynew, err = DeviceArray(info), DeviceScalar()
            lib.rkf45_attempt(info.ref, spec.ref, y.ptr, ynew.ptr, ptr_array(work), dt, err.ptr, None)
            yo_new, err_o = O.rkf45_attempt(g, orhs, to_full(grid, data), dt)
            if not np.array_equal(ynew.get_valid(), interior(grid, yo_new)) or err.value() != err_o:
                failures.append(f"{shape} {kind}: rkf45_attempt")
            lib.rhs_scaled(info.ref, spec.ref, y.ptr, ynew.ptr, 0.01, None)
            if not np.array_equal(ynew.get_valid(), interior(grid, O.rhs_scaled(g, orhs, to_full(grid, data), 0.01))):
                failures.append(f"{shape} {kind}: rhs_scaled")
    print("F32TILE " + ("OK" if not failures else "FAILED: " + "; ".join(failures)), flush=True)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
