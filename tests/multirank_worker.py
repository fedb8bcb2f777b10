"""Worker of the multi-rank tests: launched once per rank by ``python -m torch.distributed.run`` (like bench.py).

Runs the slab-parallel solves of ``CASES`` with the PRODUCT stepper (``pde_hip.distributed.SlabStepper``: one process per
device, libpdehip data plane, gloo control plane), gathers the global result on every rank and compares it on rank 0 with
the serial CPU oracle — bit for bit, equal step counts; then ``MORE``: expression conditions in the slab / block loops, block
decompositions, any expression PDE (``DecomposedExpressionStepper``) against the same library's single-device run.  Exit code
0 = all cases passed on all ranks.

Used by tests/test_hip_multirank.py (real GPUs, the real library) and by tests/test_distributed_gloo.py (the tests-only host
shim selected with PDEHIP_LIB — same worker, same loops, no GPU).
"""

from __future__ import annotations

import hashlib
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "py-pde_amd", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import pde_hip  # noqa: E402

# name: (equation, grid shape, periodic, t_range, dt (None = adaptive RKF45), solver)
CASES = {
    # one-step and two-steps-per-sweep diffusion loops (thick enough for >= 4 layers per rank at 8 ranks), fp64
    "diffusion_euler_periodic": (("diffusion", 0.8, "auto_periodic_neumann"), (64, 16, 128), [True, True, True], 1.3, 0.1, "euler"),
    "diffusion_euler_walls": (("diffusion", 0.6, {"x-": {"value": 0.3}, "x+": {"derivative": -0.2}, "y": "periodic", "z": {"value": 0.1}}),
                              (48, 8, 128), [False, True, False], 0.6, 0.05, "euler"),
    "diffusion_euler_thin": (("diffusion", 1.0, "auto_periodic_neumann"), (16, 8, 64), [True, False, True], 0.5, 0.1, "euler"),
    "diffusion_2d": (("diffusion", 1.0, "auto_periodic_neumann"), (32, 256), [True, True], 1.0, 0.1, "euler"),
    # Cahn-Hilliard: fused sweep after ONE two-layer exchange
    "cahn_hilliard_euler": (("cahn_hilliard", 0.9, "auto_periodic_neumann"), (32, 8, 128), [True, False, True], 0.01, 1e-3, "euler"),
    "cahn_hilliard_rk4": (("cahn_hilliard", 1.0, "auto_periodic_neumann"), (32, 8, 128), [False, True, True], 0.004, 1e-3, "runge-kutta"),
    # adaptive RKF45 with the MAX all-reduce and the controller in C
    "diffusion_rkf45": (("diffusion", 1.0, {"x": {"value": 0.2}, "y": "periodic", "z": "periodic"}), (32, 8, 64), [False, True, True], 1.0, None, "runge-kutta"),
    "expression_rkf45": (("expression", 1.0, "auto_periodic_neumann"), (32, 16, 64), [True, True, True], 0.05, None, "runge-kutta"),
}


# Beyond the fused slab loops - compared on rank 0 with the SAME library's single-device run (the oracle has no expression conditions /
# generic expressions; the single-device paths are pinned against the oracle and the reference elsewhere):
# name: (stepper, equation, shape, periodic, t_range, dt, solver)
_BC_EXPR = {"x-": {"value_expression": "0.2 * sin(3 * t) + 0.05 * y"}, "x+": {"derivative_expression": "0.1 * cos(t) * z - 0.3 * value**3"},
            "y": "periodic", "z-": {"virtual_point": "value / (1 + value**2) + 0.05 * x"}, "z+": {"derivative_expression": "0.05 * x * sin(t)"}}
MORE = {
    # conditions of time / position / the field, refreshed by the device program inside the slab and block C loops
    "slab_expression_bcs_rk4": ("slab", lambda: pde_hip.DiffusionPDE(0.05, bc=_BC_EXPR), (48, 8, 128), [False, True, False], 0.06, 0.01, "runge-kutta"),
    "block_expression_bcs_rkf45": ("block", lambda: pde_hip.DiffusionPDE(0.05, bc=_BC_EXPR), (32, 16, 128), [False, True, False], 0.1, None, "runge-kutta"),
    "block_cahn_hilliard_euler": ("block", lambda: pde_hip.CahnHilliardPDE(0.9), (32, 16, 128), [True, False, True], 0.005, 1e-3, "euler"),
    # the fast block loop (two steps per sweep, two-layer halos incl. edges, one message per neighbouring rank); 13 steps: 12 fast + 1
    "block_diffusion_euler_fast": ("block", lambda: pde_hip.DiffusionPDE(0.8), (32, 16, 128), [True, True, True], 1.3, 0.1, "euler"),
    "block_diffusion_euler_walls": ("block", lambda: pde_hip.DiffusionPDE(0.6, bc={"x": "periodic", "y": "periodic", "z": {"value": 0.2}}),
                                         (32, 16, 128), [True, True, False], 0.6, 0.05, "euler"),
    # any expression PDE: compiled passes per rank + ghost exchange per operator operand (slabs and the blocks of the reference's rule)
    "generic_nested_rk4": ("generic:slab", lambda: pde_hip.PDE({"c": "laplace(c**3 - c - 0.8 * laplace(c)) + 0.01 * x"}, bc={"x": {"derivative": 0}, "y": "periodic", "z": {"value": 0.1}}),
                           (32, 8, 128), [False, True, False], 0.004, 1e-3, "runge-kutta"),
    "generic_divgrad_rkf45": ("generic:auto", lambda: pde_hip.PDE({"c": "divergence((1.2 + tanh(y)) * gradient(c)) - 0.1 * c**3"}, bc={"x": "periodic", "y": {"derivative": 0.05}, "z": "periodic"}),
                              (32, 16, 128), [True, False, True], 0.2, None, "runge-kutta"),
}


# cuts along more than one axis (the reference's rule would cut these elongated grids along z only); other world sizes: that rule
BLOCKS = {2: [1, 2, 1], 4: [2, 1, 2], 8: [2, 2, 2]}


def make_eq(spec):
    kind, param, bc = spec
    if kind == "diffusion":
        return pde_hip.DiffusionPDE(param, bc=bc)
    if kind == "cahn_hilliard":
        return pde_hip.CahnHilliardPDE(param, bc_c=bc, bc_mu=bc)
    return pde_hip.PDE({"c": "laplace(c**3 - c - laplace(c))"}, bc=bc)   # BASELINE config 5


def main() -> int:
    import torch.distributed as dist

    from pde_hip.distributed import SlabStepper, TorchControl

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    control = TorchControl()
    only = set(sys.argv[1:])
    failures, report = [], {}
    for name, (spec, shape, periodic, t_range, dt, solver) in CASES.items():
        if only and name not in only:
            continue
        if shape[0] < world:
            continue
        grid = pde_hip.UnitGrid(shape, periodic=periodic)
        data = np.random.default_rng(7).uniform(-0.5, 0.5, shape)      # replicated initial state
        eq = make_eq(spec)
        st = SlabStepper(eq, grid, control=control, device=local_rank)
        final, info = st.solve(data, t_range, dt, solver)
        st.close()
        report[name] = {"steps": info["steps"], "flags": info["flags"], "two_steps_per_sweep": info["two_steps_per_sweep"]}
        if rank == 0:
            from test_oracle_golden import oracle_solve

            case = {"bc": spec[2], "t_range": t_range, "dt": dt, "solver": solver, "pde": "diffusion" if spec[0] == "diffusion" else "cahn_hilliard",
                    "D": spec[1], "gamma": spec[1]}
            expect, steps, dt_last = oracle_solve(case, grid, np.float64, data)
            if info["steps"] != steps:
                failures.append(f"{name}: {info['steps']} steps, oracle {steps}")
            elif not np.array_equal(final, expect):
                failures.append(f"{name}: max abs difference {np.abs(final - expect).max():.3e}")
            elif dt is None and abs(info["dt"] - dt_last) > 1e-12 * dt_last:
                failures.append(f"{name}: next dt {info['dt']} vs oracle {dt_last}")
        # every rank holds the same gathered field
        digests = control.allgather(hashlib.sha1(final.tobytes()).hexdigest())
        if len(set(digests)) != 1:
            failures.append(f"{name}: ranks gathered different fields")
    from pde_hip.distributed import BlockStepper, DecomposedExpressionStepper

    for name, (kind, mk_eq, shape, periodic, t_range, dt, solver) in MORE.items():
        if (only and name not in only) or shape[0] < world:
            continue
        grid = pde_hip.UnitGrid(shape, periodic=periodic)
        data = np.random.default_rng(7).uniform(-0.5, 0.5, shape)
        eq = mk_eq()
        state = pde_hip.ScalarField(grid, data)
        if kind == "slab":
            st = SlabStepper(eq, grid, control=control, device=local_rank)
        elif kind == "block":
            st = BlockStepper(eq, grid, dims=BLOCKS.get(world), control=control, device=local_rank)
        else:
            dims = kind.split(":")[1]
            st = DecomposedExpressionStepper(eq, state, dims=BLOCKS.get(world, dims) if dims == "auto" else dims, control=control, device=local_rank)
        final, info = st.solve(data, t_range, dt, solver)
        st.close()
        report[name] = {"steps": info["steps"], "decomposition": [int(d) for d in getattr(st, "dims", [world])]}
        if "fast" in name:
            report[name]["fast_block_loop"] = bool(getattr(st, "block2", False))
        if rank == 0:
            expect, sinfo = eq.solve(state, t_range, dt, solver=solver, ret_info=True)
            if info["steps"] != sinfo["solver"]["steps"]:
                failures.append(f"{name}: {info['steps']} steps, single device {sinfo['solver']['steps']}")
            elif not np.array_equal(final, expect.data):
                failures.append(f"{name}: max abs difference {np.abs(final - expect.data).max():.3e}")
            elif not np.abs(expect.data - data).max() > 1e-4:
                failures.append(f"{name}: nothing happened")
        digests = control.allgather(hashlib.sha1(final.tobytes()).hexdigest())
        if len(set(digests)) != 1:
            failures.append(f"{name}: ranks gathered different fields")
    all_failures = [f for fs in control.allgather(failures) for f in fs]
    if rank == 0:
        print("MULTIRANK " + json.dumps({"world": world, "cases": report, "failures": all_failures}), flush=True)
    dist.destroy_process_group()
    return 1 if all_failures else 0


if __name__ == "__main__":
    sys.exit(main())
