"""Golden data for the BASELINE.json configurations AT THEIR OWN SIZES, from the REFERENCE (py-pde).

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_configs.py

The fields are too large to commit (8 MB per 1024^2 state), so what is recorded per configuration is

* ``<id>/sha256``  — SHA-256 of the bytes of the final state (C order).  The reference's eager torch-CPU backend and
  the CPU oracle agree bit for bit for Euler steps of DiffusionPDE / CahnHilliardPDE in fp64
  (tests/test_oracle_golden.py), so the digest pins the oracle at full size; the HIP library is then compared with
  the oracle over the WHOLE field on the GPU (tests/test_hip_baseline_configs.py).
* ``<id>/sample``  — every ``stride``-th cell per axis of the final state (diagnostics, and the only check for cfg5,
  where the reference path — numpy backend + scipy operators, the only one that runs RKF45 without numba — is not
  bit-comparable: tolerance 1e-5 relative, fp32).
* ``<id>/steps``, ``<id>/dt_last``.

Inputs are regenerated from the seed (``np.random.default_rng(0)``; SURVEY.md §8d "synthetic inputs").
"""

from __future__ import annotations

import hashlib
import json
import sys
import time
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
    # cfg1: DiffusionPDE on UnitGrid([64, 64]) fp64, Euler dt=0.1, t_range=10 -> steppers.npz "diff64_euler_torch"
    dict(id="cfg2_diffusion_1024sq", pde="diffusion", D=1.0, bounds=[[0, 1024]] * 2, shape=[1024, 1024], periodic=[True, True],
         bc="auto_periodic_neumann", solver="euler", dt=0.1, t_range=100.0, backend="torch", vmin=0.0, vmax=1.0, stride=32),
    dict(id="cfg3_cahn_hilliard_512sq", pde="cahn_hilliard", gamma=1.0, bounds=[[0, 512]] * 2, shape=[512, 512], periodic=[False, False],
         bc="auto_periodic_neumann", solver="euler", dt=1e-3, t_range=1.0, backend="torch", vmin=0.0, vmax=1.0, stride=16),
    dict(id="cfg4_diffusion_512cube_6steps", pde="diffusion", D=1.0, bounds=[[0, 512]] * 3, shape=[512, 512, 512], periodic=[True] * 3,
         bc="auto_periodic_neumann", solver="euler", dt=0.1, t_range=0.6, backend="torch", vmin=0.0, vmax=1.0, stride=64),
    dict(id="cfg5_expression_256cube_f32_rkf45", pde="expression", rhs={"c": "laplace(c**3 - c - laplace(c))"}, bounds=[[0, 256]] * 3,
         shape=[256, 256, 256], periodic=[True] * 3, bc="auto_periodic_neumann", solver="runge-kutta", dt=1e-3, adaptive=True,
         t_range=0.25, backend="numpy", dtype="float32", vmin=-0.1, vmax=0.1, stride=32),
    # round 3: cfg3 at the step count of the published benchmark (10^4 Euler steps,
    # scripts/performance_solvers.py:53-66, :140) and cfg5 over >= 100 accepted steps
    dict(id="cfg3_cahn_hilliard_512sq_10k", pde="cahn_hilliard", gamma=1.0, bounds=[[0, 512]] * 2, shape=[512, 512], periodic=[False, False],
         bc="auto_periodic_neumann", solver="euler", dt=1e-3, t_range=10.0, backend="torch", vmin=0.0, vmax=1.0, stride=16),
    dict(id="cfg5_expression_256cube_f32_rkf45_long", pde="expression", rhs={"c": "laplace(c**3 - c - laplace(c))"}, bounds=[[0, 256]] * 3,
         shape=[256, 256, 256], periodic=[True] * 3, bc="auto_periodic_neumann", solver="runge-kutta", dt=1e-3, adaptive=True,
         t_range=2.6, backend="numpy", dtype="float32", vmin=-0.1, vmax=0.1, stride=32),
]


def initial_data(case) -> np.ndarray:
    """`ScalarField.random_uniform(grid, vmin, vmax, rng=default_rng(0))` (pde/fields/datafield_base.py:187-201)."""
    rng = np.random.default_rng(0)
    return rng.uniform(case["vmin"], case["vmax"], size=case["shape"]).astype(case.get("dtype", "float64"))


def main(only=None):
    path = HERE / "configs.npz"
    out = dict(np.load(path, allow_pickle=False)) if path.exists() else {}
    out["cases"] = json.dumps(CASES)
    for case in CASES:
        cid = case["id"]
        if only and cid not in only:
            continue
        grid = pde.CartesianGrid(case["bounds"], case["shape"], periodic=case["periodic"])
        dtype = np.dtype(case.get("dtype", "float64"))
        state = pde.ScalarField(grid, initial_data(case), dtype=dtype)
        ref_init = pde.ScalarField.random_uniform(grid, case["vmin"], case["vmax"], rng=np.random.default_rng(0), dtype=dtype)
        assert np.array_equal(ref_init.data, state.data), "initial data is not what the reference draws from the seed"
        if case["pde"] == "diffusion":
            eq = pde.DiffusionPDE(case["D"], bc=case["bc"])
        elif case["pde"] == "cahn_hilliard":
            eq = pde.CahnHilliardPDE(case["gamma"], bc_c=case["bc"], bc_mu=case["bc"])
        else:
            # the reference's expression class needs numba on the numpy backend (pde/pdes/pde.py:473); CahnHilliardPDE is the
            # bit-identical twin of `laplace(c**3 - c - laplace(c))`
            eq = pde.CahnHilliardPDE(1.0, bc_c=case["bc"], bc_mu=case["bc"])
        config["default_backend"] = "scipy" if case["backend"] == "numpy" else "numba"
        t0 = time.time()
        kwargs = {"adaptive": True} if case.get("adaptive") else {}
        res, info = eq.solve(state, t_range=case["t_range"], dt=case["dt"], backend=case["backend"], solver=case["solver"], tracker=None,
                             ret_info=True, **kwargs)
        s = case["stride"]
        data = np.ascontiguousarray(res.data)
        out[f"{cid}/sha256"] = np.array(hashlib.sha256(data.tobytes()).hexdigest())
        out[f"{cid}/sample"] = data[(slice(None, None, s),) * grid.num_axes].copy()
        out[f"{cid}/steps"] = np.array(info["solver"]["steps"])
        out[f"{cid}/dt_last"] = np.array(info["solver"]["dt"], dtype=np.float64)
        print(f"[{cid}] steps={info['solver']['steps']} dt_last={info['solver']['dt']} {time.time() - t0:.1f}s sha={str(out[f'{cid}/sha256'])[:16]}", flush=True)
        np.savez_compressed(path, **out)
    config["default_backend"] = "numba"


if __name__ == "__main__":
    main(set(sys.argv[1:]) or None)
