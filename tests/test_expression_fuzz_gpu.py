"""Differential fuzz on the device: random expression right-hand sides evaluated by the run-time built HIP kernels and - same
plan, same generated epilogue text - by the tests-only host shim (the CPU oracle's stencils + gcc for the epilogue).  Both sides
round every pass to the storage type and use no contraction, so fp64 results agree bit for bit; this covers the kernel side of
what tests/test_expression_fuzz.py checks against the reference on the host (vector-tile kernels, ragged rows, 1-D, the per-axis
derivative family, on-the-fly boundary conditions, pass splitting)."""

from __future__ import annotations

import numpy as np
import pytest

import pde_hip


def _random_scalar(rng, depth: int, fields: list[str], axes: str) -> str:
    f = lambda: fields[rng.integers(len(fields))]  # noqa: E731
    ax = lambda: axes[rng.integers(len(axes))]  # noqa: E731
    leaves = [lambda: f(), lambda: f"{rng.uniform(0.2, 1.5):.3f}", lambda: ax(), lambda: f"laplace({f()})", lambda: f"gradient_squared({f()})",
              lambda: f"d_d{ax()}({f()})", lambda: f"d2_d{ax()}2({f()})", lambda: f"{f()}**3"]
    if depth <= 0:
        return leaves[rng.integers(len(leaves))]()
    kind = rng.integers(9)
    a, b = _random_scalar(rng, depth - 1, fields, axes), _random_scalar(rng, depth - 1, fields, axes)
    if kind == 0:
        return f"({a} + {b})"
    if kind == 1:
        return f"({a} - {b})"
    if kind == 2:
        return f"({a} * {b})"
    if kind == 3:
        return f"laplace({a})"
    if kind == 4:
        return f"d_d{ax()}({a})"
    if kind == 5:
        return f"dot(gradient({f()}), gradient({f()}))"
    if kind == 6:
        return f"divergence(({a}) * gradient({f()}))"
    if kind == 7:
        return f"gradient_squared({a})"
    return f"({a})**2"


GRIDS = [((37,), [False]), ((64,), [True]), ((12, 130), [False, True]), ((24, 72), [True, False]), ((5, 6, 136), [False, True, True]),
         ((4, 10, 128), [True, True, False])]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,dtype", [(s, np.float64) for s in range(36)] + [(s, np.float32) for s in range(100, 112)])
def test_random_expressions_device_vs_oracle_shim(seed, dtype):
    import shimlib

    rng = np.random.default_rng(2000 + seed)
    shape, periodic = GRIDS[seed % len(GRIDS)]
    nd = len(shape)
    axes = "xyz"[:nd]
    grid = pde_hip.CartesianGrid([[0, 0.5 * n] for n in shape], shape, periodic=periodic)
    bc = {a: ("periodic" if p else ({"value": 0.2} if seed % 2 else {"derivative": -0.1})) for a, p in zip(axes, periodic)}
    two = seed % 4 == 0
    fields = ["u", "v"] if two else ["u"]
    rhs = {name: _random_scalar(rng, 1 + seed % 3, fields, axes) for name in fields}
    data = rng.uniform(-0.4, 0.4, (len(fields), *shape))

    def evaluate():
        b = pde_hip.get_backend("hip")
        if two:
            state = pde_hip.FieldCollection([pde_hip.ScalarField(grid, d, dtype=dtype) for d in data])
        else:
            state = pde_hip.ScalarField(grid, data[0], dtype=dtype)
        eq = pde_hip.PDE(rhs, bc=bc)
        rate = b.native_to_numpy(eq.make_pde_rhs(state)(b.numpy_to_native(state.data, grid=grid), 0.0))
        out = eq.solve(state, t_range=3e-4, dt=1e-4, solver="euler", backend="hip")
        return np.array(rate), np.array(out.data)

    rate_dev, out_dev = evaluate()
    with shimlib.use_shim(fused=False):
        rate_ref, out_ref = evaluate()
    assert np.isfinite(rate_ref).all(), rhs
    if dtype == np.float64:
        np.testing.assert_array_equal(rate_dev, rate_ref, err_msg=str(rhs))
        np.testing.assert_array_equal(out_dev, out_ref, err_msg=str(rhs))
    else:
        # fp32 storage: the kernels keep the stencil values in fp64 registers, the shim rounds them to fp32 between the
        # oracle's operator and the epilogue - agreement to fp32 rounding of the intermediate terms
        from helpers import max_rel

        assert rate_dev.dtype == np.float32 and max_rel(rate_dev.astype(np.float64), rate_ref.astype(np.float64)) < 2e-5, rhs
        assert max_rel(out_dev.astype(np.float64), out_ref.astype(np.float64)) < 2e-5, rhs
