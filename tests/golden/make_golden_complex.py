"""Golden runs of COMPLEX-valued equations by the reference (run in the build container; needs /root/reference).

`pde.PDE` evaluates expressions through numba on the numpy backend, which is not installable here; the right-hand sides are therefore
restated with the reference's own field operators inside `PDEBase.evolution_rate` (complex arrays through its scipy operators) and
solved by the reference's solvers (Euler, RK4, adaptive RKF45, adaptive Euler - the error norm of a complex state is the modulus,
pde/solvers/runge_kutta.py:147-148).  tests/test_hip_complex.py rebuilds the equations as `pde_hip.PDE` expressions on the GPU.

    python tests/golden/make_golden_complex.py   ->  tests/golden/complex.npz
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

G = 0.3 - 0.8j
CASES = [
    dict(id="schroedinger_2d", shape=[8, 6], periodic=[False, True], rhs="I * laplace(p)", var="p", bc={"x": {"value": [1.0, 2.0]}, "y": "periodic"},
         t_range=0.05, dt=1e-3),
    dict(id="gross_pitaevskii_2d", shape=[10, 8], periodic=[True, False], rhs="-I * laplace(c) + g * c * Abs(c)**2 - 0.1 * conjugate(c)", var="c",
         consts={"g": [G.real, G.imag]}, bc={"x": "periodic", "y": {"derivative": [0.1, -0.2]}}, t_range=0.05, dt=1e-3),
    dict(id="schroedinger_3d", shape=[12, 10, 64], periodic=[True, False, True], rhs="(0.2 + I) * laplace(p)", var="p",
         bc={"x": "periodic", "y": {"derivative": [0.05, 0.1]}, "z": "periodic"}, t_range=0.04, dt=2e-3),
    # round 5: conditions given as EXPRESSIONS of time and position with complex values (the factor of `value` is real: the parts decouple)
    dict(id="gross_pitaevskii_expression_bcs_2d", shape=[8, 6], periodic=[False, True], rhs="I * laplace(p) - 0.1 * p * Abs(p)**2", var="p",
         bc={"x-": {"value_expression": "I*t + 0.1*y"}, "x+": {"derivative_expression": "(1 + 2*I)*cos(t) - 0.5*value"}, "y": "periodic"},
         t_range=0.05, dt=1e-3),
    # round 5: vector operators of complex arguments (`dot` conjugates its second operand, pde/fields/datafield_base.py:965-986)
    dict(id="divgrad_dot_2d", shape=[8, 6], periodic=[True, False], var="c",
         rhs="I * divergence((1 + 0.5*I) * gradient(c)) + 0.1 * dot(gradient(c), gradient(c)) - 0.1 * c",
         bc={"x": "periodic", "y": {"value": [0.3, -0.2]}}, t_range=0.02, dt=1e-3),
    # round 6: mixed (Robin) conditions with a COMPLEX coefficient of the field value (pde/grids/boundaries/local.py:1927-1938): the factor of the
    # virtual point is complex, real and imaginary part of the ghost cells depend on both parts of the field
    dict(id="schroedinger_robin_2d", shape=[10, 8], periodic=[False, True], rhs="(0.2 + I) * laplace(p)", var="p",
         bc={"x-": {"type": "mixed", "value": [0.5, 1.5], "const": [0.2, -0.3]}, "x+": {"value": [1.0, 2.0]}, "y": "periodic"}, t_range=0.05, dt=1e-3),
    dict(id="expression_complex_slope_2d", shape=[8, 6], periodic=[False, True], rhs="I * laplace(p) - 0.1 * p * Abs(p)**2", var="p",
         bc={"x-": {"value_expression": "I*t + 0.1*y"}, "x+": {"derivative_expression": "(1 + 2*I)*cos(t) - (0.5 + 0.3*I)*value"}, "y": "periodic"},
         t_range=0.05, dt=1e-3),
    dict(id="schroedinger_robin_3d", shape=[6, 8, 64], periodic=[False, False, True], rhs="(0.2 + I) * laplace(p)", var="p",
         bc={"x": {"type": "mixed", "value": [-0.4, 0.8], "const": [0.1, 0.2]}, "y-": {"derivative": [0.05, 0.1]},
             "y+": {"type": "mixed", "value": [0.3, -0.6], "const": 0.0}, "z": "periodic"}, t_range=0.04, dt=2e-3),
]
SOLVERS = [("euler", False), ("runge-kutta", False), ("runge-kutta", True), ("euler", True)]


def cplx(v):
    return complex(v[0], v[1]) if isinstance(v, list) else v


def bc_of(case):
    return {k: ({kk: cplx(vv) for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in case["bc"].items()}


def main():
    rng = np.random.default_rng(21)
    out = {"cases": json.dumps(CASES), "solvers": json.dumps(SOLVERS)}
    for case in CASES:
        grid = pde.UnitGrid(case["shape"], periodic=case["periodic"])
        bc = bc_of(case)
        cid = case["id"]

        class Eq(pde.PDEBase):
            complex_valued = True

            def evolution_rate(self, state, t=0, cid=cid, bc=bc):
                if cid in ("gross_pitaevskii_expression_bcs_2d", "expression_complex_slope_2d"):
                    c = state.data
                    return pde.ScalarField(state.grid, 1j * state.laplace(bc, args={"t": t}).data - 0.1 * c * np.abs(c) ** 2)
                if cid == "divgrad_dot_2d":
                    grad = state.gradient(bc)
                    div = ((1 + 0.5j) * grad).divergence(bc)
                    dot = np.einsum("i...,i...->...", grad.data, grad.data.conjugate())
                    return pde.ScalarField(state.grid, 1j * div.data + 0.1 * dot - 0.1 * state.data)
                c, lap = state.data, state.laplace(bc).data
                if cid == "schroedinger_2d":
                    rate = 1j * lap
                elif cid == "gross_pitaevskii_2d":
                    rate = -1j * lap + G * c * np.abs(c) ** 2 - 0.1 * np.conjugate(c)
                else:
                    rate = (0.2 + 1j) * lap
                return pde.ScalarField(state.grid, rate)

        init = rng.uniform(-0.5, 0.5, grid.shape) + 1j * rng.uniform(-0.5, 0.5, grid.shape)
        out[f"{cid}/input"] = init
        for solver, adaptive in SOLVERS:
            res, info = Eq().solve(pde.ScalarField(grid, init), t_range=case["t_range"], dt=case["dt"], solver=solver, adaptive=adaptive, tracker=None,
                                   ret_info=True, backend="numpy")
            key = f"{cid}/{solver}{'_adaptive' if adaptive else ''}"
            out[f"{key}/final"] = res.data.copy()
            out[f"{key}/steps"] = np.array(info["solver"]["steps"])
            print(key, "steps", info["solver"]["steps"], "max |u|", float(np.abs(res.data).max()))
    np.savez_compressed(HERE / "complex.npz", **out)


if __name__ == "__main__":
    main()
