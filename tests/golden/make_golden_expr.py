"""Golden vectors for generic expression PDEs (run in the build container; needs /root/reference).

The reference's eager torch-CPU backend evaluates `PDE({...})` expressions (operators substituted into
the sympy expression, pde/pdes/pde.py:299-499) — the only reference backend that can do so without
numba.  Recorded: the evolution rate of a random state and the state after a short fixed-step Euler
run.  The hip backend evaluates the same expressions with run-time specialised kernels; since sympy
may order commutative terms differently the comparison tolerance is 1e-10 relative, not bit-exact.
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
from pde import config  # noqa: E402

config["backend.torch.compile"] = False
HERE = Path(__file__).resolve().parent

CASES = [
    dict(id="allen_cahn_2d", rhs={"c": "c - c**3 + laplace(c)"}, consts={}, bounds=[[0, 8], [0, 8]], shape=[16, 16], periodic=[True, False],
         bc="auto_periodic_neumann", dt=0.01, t_range=0.5),
    dict(id="kpz_2d", rhs={"h": "nu * laplace(h) + lam * gradient_squared(h)"}, consts={"nu": 0.5, "lam": 1.5}, bounds=[[0, 8], [0, 6]], shape=[16, 12],
         periodic=[True, True], bc="auto_periodic_neumann", dt=0.01, t_range=0.3),
    dict(id="swift_hohenberg_2d", rhs={"c": "(eps - kc2**2) * c - 2 * kc2 * laplace(c) - laplace(laplace(c)) + delta * c**2 - c**3"},
         consts={"eps": 0.1, "kc2": 1.0, "delta": 0.2}, bounds=[[0, 16], [0, 16]], shape=[16, 16], periodic=[True, True], bc="auto_periodic_neumann",
         dt=0.005, t_range=0.1),
    dict(id="ks_1d", rhs={"u": "-laplace(laplace(u)) - laplace(u) - 0.5 * gradient_squared(u)"}, consts={}, bounds=[[0, 32]], shape=[32], periodic=[True],
         bc="auto_periodic_neumann", dt=0.002, t_range=0.05),
    dict(id="reaction_3d_dirichlet", rhs={"c": "D * laplace(c) + c * (1 - c) - 0.1 * c**4"}, consts={"D": 0.3}, bounds=[[0, 4], [0, 4], [0, 8]], shape=[6, 6, 8],
         periodic=[False, True, False], bc={"x": {"value": 0.5}, "y": "periodic", "z": {"derivative": 0.1}}, dt=0.01, t_range=0.2),
    dict(id="nested_nonlinear_2d", rhs={"c": "laplace(c**3 - c - 0.7 * laplace(c)) + 0.1 * c"}, consts={}, bounds=[[0, 8], [0, 8]], shape=[16, 16], periodic=[True, False],
         bc="auto_periodic_neumann", dt=1e-3, t_range=0.05),
]


def main():
    rng = np.random.default_rng(3)
    out = {"cases": json.dumps(CASES)}
    for case in CASES:
        cid = case["id"]
        grid = pde.CartesianGrid(case["bounds"], case["shape"], periodic=case["periodic"])
        state = pde.ScalarField.random_uniform(grid, -0.5, 0.5, rng=rng)
        eq = pde.PDE(case["rhs"], bc=case["bc"], consts=case["consts"])
        rhs = eq.make_pde_rhs(state, backend="torch")
        import torch

        rate = rhs(torch.from_numpy(np.ascontiguousarray(state.data)), 0.0)
        res, info = eq.solve(state, t_range=case["t_range"], dt=case["dt"], backend="torch", solver="euler", tracker=None, ret_info=True)
        out[f"{cid}/input"] = state.data.copy()
        out[f"{cid}/rate"] = np.asarray(rate)
        out[f"{cid}/final"] = res.data.copy()
        out[f"{cid}/steps"] = np.array(info["solver"]["steps"])
        print(cid, "steps", info["solver"]["steps"], "max|rate|", float(np.abs(np.asarray(rate)).max()))
    np.savez_compressed(HERE / "exprs.npz", **out)


if __name__ == "__main__":
    main()
