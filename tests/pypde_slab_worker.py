"""One rank of a slab-parallel run driven by the REAL py-pde (launched by torch.distributed.run; CPU: host shim via PDEHIP_LIB).

`eq.solve(state, ..., solver="hip_slab", backend="hip")` — pde_hip.pypde_plugin.HipSlabSolver, the counterpart of the
reference's ExplicitMPISolver — on every rank; rank 0 compares with the reference's own serial numpy + scipy run.
"""

from __future__ import annotations

import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "py-pde_amd", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))
from refpath import add_to_path  # noqa: E402

add_to_path()


def main() -> int:
    import torch.distributed as dist

    import pde

    import pde_hip.pypde_plugin  # noqa: F401

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    failures, report = [], {}
    cases = {
        "diffusion_euler": (pde.DiffusionPDE(0.7, bc={"x": {"value": 0.2}, "y": "periodic", "z": {"derivative": 0.1}}),
                            pde.UnitGrid([12, 4, 6], periodic=[False, True, False]), dict(t_range=0.5, dt=0.05)),
        "cahn_hilliard_rk4": (pde.CahnHilliardPDE(0.9), pde.UnitGrid([8, 4, 6], periodic=True), dict(t_range=0.004, dt=1e-3, scheme="runge-kutta")),
        # conditions that depend on time, on the position along decomposed and undecomposed axes, and non-linearly on the field
        "diffusion_expression_bcs": (pde.DiffusionPDE(0.6, bc={"x-": {"value_expression": "0.2 * sin(3 * t) + 0.05 * y"},
                                                               "x+": {"derivative_expression": "-0.3 * value**3 + 0.02 * z"}, "y": "periodic",
                                                               "z-": {"derivative_expression": "0.05 * x * sin(t)"}, "z+": {"value": 0.1}}),
                                     pde.UnitGrid([12, 4, 6], periodic=[False, True, False]), dict(t_range=0.4, dt=0.02, scheme="runge-kutta")),
        "expression_rkf45": (pde.PDE({"c": "laplace(c**3 - c - laplace(c))"}), pde.UnitGrid([8, 6, 6], periodic=True),
                             dict(t_range=0.1, dt=1e-3, scheme="runge-kutta", adaptive=True)),
    }
    # PDEs WITHOUT a fused decomposed loop: the run-time compiled passes on the box of each rank, a ghost exchange before every pass that
    # applies operators (pde_hip.distributed.DecomposedExpressionStepper).  Yardsticks: the reference's numpy backend for its classes, its
    # eager torch-CPU backend for `pde.PDE` (Euler only; that class takes its operators from numba on the numpy backend)
    wall = {"x": {"value": 0.2}, "y": "periodic", "z": {"derivative": 0.1}}
    cases.update({
        "allen_cahn_class_rkf45": (pde.AllenCahnPDE(0.9, mobility=1.1, bc=wall), pde.UnitGrid([12, 4, 6], periodic=[False, True, False]),
                                   dict(t_range=0.3, dt=1e-2, scheme="runge-kutta", adaptive=True)),
        "diffusion_adaptive_euler": (pde.DiffusionPDE(0.6, bc=wall), pde.UnitGrid([12, 4, 6], periodic=[False, True, False]),
                                     dict(t_range=0.5, dt=1e-2, adaptive=True)),
        # the reference's adaptive Euler loop (rate carried between attempts, evaluated at the old time) with conditions that depend on time,
        # on the decomposed and on an undecomposed axis: the fused slab / block loop (Diffusion, Cahn-Hilliard) and the decomposed passes
        "diffusion_adaptive_euler_time_bcs": (pde.DiffusionPDE(0.6, bc={"x-": {"value_expression": "0.2 * sin(3 * t) + 0.05 * y"}, "x+": {"derivative": 0.1},
                                                                        "y": "periodic", "z-": {"derivative_expression": "0.05 * x * sin(t)"}, "z+": {"value": 0.1}}),
                                              pde.UnitGrid([12, 4, 6], periodic=[False, True, False]), dict(t_range=0.5, dt=0.2, adaptive=True)),
        "cahn_hilliard_adaptive_euler_time_bcs": (pde.CahnHilliardPDE(0.9, bc_c={"x-": {"value_expression": "0.1 * cos(2 * t)"}, "x+": {"derivative": 0}, "y": "periodic"}),
                                                  pde.UnitGrid([12, 8], periodic=[False, True]), dict(t_range=0.05, dt=1e-3, adaptive=True)),
        "allen_cahn_adaptive_euler_time_bcs": (pde.AllenCahnPDE(0.9, bc={"x-": {"value_expression": "0.2 * sin(3 * t)"}, "x+": {"derivative": 0.1}, "y": "periodic"}),
                                               pde.UnitGrid([12, 8], periodic=[False, True]), dict(t_range=0.3, dt=0.1, adaptive=True)),
        "swift_hohenberg_class_rk4": (pde.SwiftHohenbergPDE(rate=0.1, kc2=0.8, delta=0.3), pde.UnitGrid([10, 8], periodic=[True, False]),
                                      dict(t_range=0.008, dt=1e-3, scheme="runge-kutta")),
        "pde_nested_euler": (pde.PDE({"c": "laplace(c**3 - c - 0.7 * laplace(c)) + 0.01 * x * y"}, bc={"x": {"derivative": 0.05}, "y": "periodic"}),
                             pde.CartesianGrid([[0, 10], [0, 8]], [10, 8], periodic=[False, True]), dict(t_range=0.02, dt=1e-3), {"ref": "torch"}),
        "pde_brusselator_euler": (pde.PDE({"u": "laplace(u) + 1 - 3 * u + u**2 * v", "v": "0.1 * laplace(v) + 2 * u - u**2 * v"},
                                          bc={"x": "periodic", "y": {"derivative": 0.1}}),
                                  pde.UnitGrid([12, 8], periodic=[True, False]), dict(t_range=0.05, dt=2e-3), {"ref": "torch", "fields": 2}),
        # the scenarios of the reference's own MPI tests (tests/solvers/test_explicit_mpi_solvers.py:22-53, :87-115): adaptive Euler with the
        # decomposition [1, -1], two coupled equations with adaptive steps, noise (below)
        "ref_simple_adaptive": (pde.DiffusionPDE(), pde.UnitGrid([8, 8], periodic=[True, False]), dict(t_range=1.01, dt=0.1, adaptive=True),
                                {"decomposition": [1, -1]}),
        "ref_multiple_pdes": (pde.PDE({"a": "laplace(a) - b", "b": "laplace(b) + a"}), pde.UnitGrid([8, 8], periodic=[True, False]),
                              dict(t_range=1.01, dt=0.1, adaptive=True), {"ref": "torch", "fields": 2, "tol": 1e-5}),
        # (the torch yardstick's adaptive loop ends 1e-6 past t_range - its mean step times 104 steps is 1.010001 - where the numpy / numba loops
        # and this backend clip the last step to t_end, pde/backends/numba/_solvers.py:249-281; the serial hip run differs by the same 1.0e-6)
        "pde_vector_euler": (pde.PDE({"u": "vector_laplace(u) - u + 0.1 * gradient(dot(u, u))"}, bc={"x": "periodic", "y": {"derivative": 0.1}}),
                             pde.UnitGrid([12, 8], periodic=[True, False]), dict(t_range=0.05, dt=2e-3), {"ref": "torch", "vector": True}),
    })
    # differential fuzz (PDEHIP_WORKER_FUZZ=n): random grids, random conditions per face - constants, mixed, curvature, expressions
    # of time and position, expressions that read the field -, random solver; every rank draws the same case from the same seed
    for k in range(int(os.environ.get("PDEHIP_WORKER_FUZZ", "0"))):
        rng = np.random.default_rng(9000 + k)
        nd = 1 + k % 3
        shape = [int(rng.integers(max(5, 2 * world), 14))] + [int(rng.integers(4, 13)) for _ in range(nd - 1)]
        periodic = [bool(rng.integers(2)) for _ in range(nd)]
        dx = float(rng.choice([0.5, 1.0, 2.0]))
        grid = pde.CartesianGrid([[0, dx * n] for n in shape], shape, periodic=periodic)

        def face(axes, nonlinear, rng=rng):
            kind = int(rng.integers(8 if nonlinear else 6)) if axes else int(rng.integers(4))
            a, b = float(rng.uniform(0.1, 0.5)), float(rng.uniform(-0.2, 0.2))
            pick = axes[int(rng.integers(len(axes)))] if axes else ""
            return [{"value": b}, {"derivative": b}, {"type": "mixed", "value": a, "const": b}, {"curvature": b},
                    {"value_expression": f"0.1 * sin(3 * t) + 0.05 * {pick}"}, {"derivative_expression": f"0.1 * cos(t) * {pick}"},
                    {"derivative_expression": f"-{a:.3f} * value**3 + {b:.3f} * {pick}"}, {"value_expression": f"{a:.3f} * tanh(value) + {b:.3f} * sin(2 * t)"}][kind]

        def conditions(nonlinear, grid=grid):
            bc = {}
            for ax, per in zip(grid.axes, grid.periodic):
                others = "".join(a for a in grid.axes if a != ax)
                if per:
                    bc[ax] = "periodic"
                else:
                    bc[ax + "-"], bc[ax + "+"] = face(others, nonlinear), face(others, nonlinear)
            return bc

        if k % 2 == 0:
            eq = pde.DiffusionPDE(float(rng.uniform(0.2, 1.5)), bc=conditions(True))
        else:
            eq = pde.CahnHilliardPDE(float(rng.uniform(0.5, 1.5)), bc_c=conditions(True), bc_mu=conditions(False))
        dt = 1e-3 * dx**4
        kw = dict(t_range=8 * dt, dt=dt)
        if rng.integers(2):
            kw["scheme"] = "runge-kutta"
            if k % 3 == 0:
                kw["adaptive"] = True
        elif k % 3 == 1:
            kw["adaptive"] = True      # the reference's adaptive Euler loop
        cases[f"fuzz{k}"] = (eq, grid, kw)
    pde.config["default_backend"] = "scipy"
    for name, (eq, grid, kw, *rest) in cases.items():
        opts = rest[0] if rest else {}
        state = pde.ScalarField.random_uniform(grid, -0.4, 0.4, rng=np.random.default_rng(3))
        if opts.get("vector"):
            state = pde.VectorField.random_uniform(grid, -0.4, 0.4, rng=np.random.default_rng(3))
        if opts.get("fields"):
            state = pde.FieldCollection([pde.ScalarField.random_uniform(grid, 0.1, 0.9, rng=np.random.default_rng(3 + k)) for k in range(opts["fields"])])
        seen = []
        tracker = pde.CallbackTracker(lambda s, t: seen.append((t, float(s.data.sum()))), interrupts=kw["t_range"] / 2)
        decomposition = os.environ.get("PDEHIP_WORKER_DECOMPOSITION", "slab")    # "auto": blocks by the reference's rule
        # PDEHIP_WORKER_GATHER=root: the field goes to rank 0 only when a stepper call ends (the other ranks keep their own part)
        gather = os.environ.get("PDEHIP_WORKER_GATHER", "all")
        res, info = eq.solve(state, solver="hip_slab", backend="hip", tracker=tracker, ret_info=True,
                             decomposition=opts.get("decomposition", decomposition), gather=gather, **kw)
        if opts.get("decomposition") and world > 1 and info["solver"]["decomposition"] != [1, world]:
            failures.append(f"{name}: decomposition {info['solver']['decomposition']}")
        report[name] = {"steps": info["solver"]["steps"], "world": info["solver"]["world_size"], "interrupts": len(seen),
                        "decomposition": info["solver"]["decomposition"]}
        if rank == 0:
            ref_kw = {k: v for k, v in kw.items() if k != "scheme"}
            ref_eq = pde.CahnHilliardPDE() if name == "expression_rkf45" else eq   # the expression class needs numba on numpy
            pde.config["backend.torch.compile"] = False
            ref, rinfo = ref_eq.solve(state, solver="runge-kutta" if kw.get("scheme") else "euler", backend=opts.get("ref", "numpy"),
                                      tracker=pde.CallbackTracker(lambda s, t: None, interrupts=kw["t_range"] / 2), ret_info=True, **ref_kw)
            err = np.abs(res.data - ref.data).max() / np.abs(ref.data).max()
            if info["solver"]["steps"] != rinfo["solver"]["steps"]:
                failures.append(f"{name}: {info['solver']['steps']} steps, reference {rinfo['solver']['steps']}")
            if not err < opts.get("tol", 1e-9 if name.startswith("fuzz") else 1e-10):
                failures.append(f"{name}: relative difference {err:.3e}")
            if len(seen) != 3:
                failures.append(f"{name}: {len(seen)} tracker interrupts")
    # post-step hooks act on the box of each rank, like under the reference's MPI solver (pde/solvers/explicit_mpi.py:43-49: "local
    # modifications"); a pointwise hook therefore equals its serial application.  StopIteration on one rank ends the run on all of them.
    class Clipped(pde.DiffusionPDE):
        def __init__(self, stop_at=None, **kw):
            super().__init__(**kw)
            self.stop_at = stop_at

        def make_post_step_hook(self, state, backend="numpy"):
            stop_at = self.stop_at

            def hook(state_data, t, post_step_data):
                np.minimum(state_data, 0.25, out=state_data)
                if stop_at is not None and t > stop_at and state_data.max() > -1:
                    raise StopIteration
                post_step_data += 1
                return state_data, post_step_data

            return hook, 0

    grid = pde.UnitGrid([12, 4, 6], periodic=[False, True, False])
    state = pde.ScalarField.random_uniform(grid, -0.4, 0.4, rng=np.random.default_rng(3))
    for label, stop_at in (("hook", None), ("hook_stop", 0.12)):
        eq = Clipped(stop_at, diffusivity=0.6, bc={"x": {"value": 0.2}, "y": "periodic", "z": {"derivative": 0.1}})
        kw = dict(t_range=0.3, dt=0.02, tracker=None, ret_info=True)
        res, info = eq.solve(state, solver="hip_slab", backend="hip", decomposition=os.environ.get("PDEHIP_WORKER_DECOMPOSITION", "slab"), **kw)
        ref, rinfo = eq.solve(state, solver="euler", backend="numpy", **kw)
        report[label] = {"steps": info["solver"]["steps"], "world": world, "interrupts": 3, "decomposition": info["solver"]["decomposition"]}
        err = np.abs(res.data - ref.data).max() / np.abs(ref.data).max()
        # (a run ended by the hook: the reference's numpy loop leaves `steps` at 0 - it counts after the loop -, this backend counts the
        # steps it took)
        if not err < 1e-10 or (stop_at is None and info["solver"]["steps"] != rinfo["solver"]["steps"]):
            failures.append(f"{label}: relative difference {err:.3e}, steps {info['solver']['steps']} / {rinfo['solver']['steps']}")
        if info["solver"]["post_step_data"] != rinfo["solver"]["post_step_data"] or info["solver"]["post_step_data_list"] != [rinfo["solver"]["post_step_data"]] * world:
            failures.append(f"{label}: hook data {info['solver']['post_step_data']} {info['solver'].get('post_step_data_list')} / {rinfo['solver']['post_step_data']}")
        if info["controller"]["t_final"] != rinfo["controller"]["t_final"] or not res.data.max() <= 0.25:
            failures.append(f"{label}: t_final {info['controller']['t_final']} / {rinfo['controller']['t_final']}")
    # noise (tests/solvers/test_explicit_mpi_solvers.py:56-82): a tiny variance leaves the deterministic result, `info` says stochastic
    field = pde.ScalarField.random_uniform(pde.UnitGrid([16]), -1, 1, rng=np.random.default_rng(5))
    runs = {}
    for label, eq in (("plain", pde.DiffusionPDE()), ("noisy", pde.DiffusionPDE(noise=1e-10)), ("loud", pde.DiffusionPDE(diffusivity=0.0, noise=0.5))):
        out, info = eq.solve(field, t_range=0.2 if label != "loud" else 1e-3, dt=1e-3, solver="hip_slab", backend="hip", tracker=None, ret_info=True)
        runs[label] = (out.data.copy(), info["solver"])
    report["stochastic"] = {"steps": runs["noisy"][1]["steps"], "world": world, "interrupts": 3, "decomposition": runs["noisy"][1]["decomposition"]}
    if not np.allclose(runs["plain"][0], runs["noisy"][0], rtol=1e-4, atol=1e-4) or np.array_equal(runs["plain"][0], runs["noisy"][0]):
        failures.append("stochastic: tiny noise must leave the deterministic result (and still change it)")
    if runs["plain"][1]["stochastic"] or not runs["noisy"][1]["stochastic"] or runs["noisy"][1]["dt_adaptive"]:
        failures.append(f"stochastic: info {runs['plain'][1]['stochastic']} {runs['noisy'][1]['stochastic']}")
    # one step of pure noise: increments sqrt(dt * noise / V) * N(0, 1), independent on every box (no two boxes share their numbers)
    inc = (runs["loud"][0] - field.data) / np.sqrt(1e-3 * 0.5)
    if not 0.6 < inc.std() < 1.4 or len(np.unique(np.round(inc, 12))) != inc.size:
        failures.append(f"stochastic: increments std {inc.std():.3f}, {len(np.unique(np.round(inc, 12)))} distinct of {inc.size}")
    if rank == 0:
        print("PYPDESLAB " + json.dumps({"world": world, "cases": report, "failures": failures}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
