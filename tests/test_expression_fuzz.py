"""Differential fuzz of the expression compiler: random right-hand sides built from the supported operators, evaluated by the
hip backend (tests-only host shim: exercises parsing, vector lowering, pass planning / splitting, code generation, BC tables)
and by the REFERENCE's eager torch-CPU backend on the same `pde.PDE` object.  Seeds are fixed: the cases are reproducible."""

from __future__ import annotations

import numpy as np
import pytest

import sys
from pathlib import Path

from refpath import REF  # noqa: E402
if not (REF / "pde").exists():
    pytest.skip("py-pde (reference) not available", allow_module_level=True)
if str(REF) not in sys.path:
    sys.path.append(str(REF))

import pde  # noqa: E402
import torch  # noqa: E402

import shimlib  # noqa: E402
from helpers import max_rel  # noqa: E402


def _random_scalar(rng, depth: int, fields: list[str]) -> str:
    f = lambda: fields[rng.integers(len(fields))]  # noqa: E731
    # (no d_dx family here: the reference has it in its numba backend only; tests/test_expressions.py checks it against the oracle)
    leaves = [lambda: f(), lambda: f"{rng.uniform(0.2, 1.5):.3f}", lambda: "x", lambda: "y", lambda: f"laplace({f()})",
              lambda: f"gradient_squared({f()})", lambda: f"{f()}**3"]
    if depth <= 0:
        return leaves[rng.integers(len(leaves))]()
    kind = rng.integers(9)
    a, b = _random_scalar(rng, depth - 1, fields), _random_scalar(rng, depth - 1, fields)
    if kind == 0:
        return f"({a} + {b})"
    if kind == 1:
        return f"({a} - {b})"
    if kind == 2:
        return f"({a} * {b})"
    if kind == 3:
        return f"laplace({a})"
    if kind == 4:
        return f"({a})**3"   # (no elementary functions: the reference's torch path cannot apply them to tensors)
    if kind == 5:
        return f"dot(gradient({f()}), gradient({f()}))"
    if kind == 6:
        return f"divergence(({a}) * gradient({f()}))"
    if kind == 7:
        return f"gradient_squared({a})"
    return f"({a})**2"


@pytest.mark.parametrize("seed", range(40))
def test_random_expressions_match_the_reference(seed, monkeypatch):
    monkeypatch.setitem(pde.config, "backend.torch.compile", False)
    rng = np.random.default_rng(1000 + seed)
    grid = pde.CartesianGrid([[0, 3], [-1, 1]], [12, 10], periodic=[bool(seed % 2), True])
    bc = {"x": "periodic" if seed % 2 else {"value": 0.2}, "y": "periodic"}
    two = seed % 3 == 0
    fields = ["u", "v"] if two else ["u"]
    rhs = {name: _random_scalar(rng, 2 + seed % 2, fields) for name in fields}
    data = rng.uniform(-0.4, 0.4, (len(fields), *grid.shape))
    state = pde.FieldCollection([pde.ScalarField(grid, d) for d in data]) if two else pde.ScalarField(grid, data[0])
    eq = pde.PDE(rhs, bc=bc)
    try:
        expect = eq.make_pde_rhs(state, backend="torch")(torch.from_numpy(np.ascontiguousarray(state.data)), 0.0)
    except Exception as err:   # noqa: BLE001 - the reference's torch path refuses some forms (e.g. a bare constant)
        pytest.skip(f"reference cannot evaluate {rhs}: {type(err).__name__}")
    expect = np.asarray(expect)
    if expect.shape != state.data.shape or not np.isfinite(expect).all():
        pytest.skip(f"reference result unusable for {rhs}")
    with shimlib.use_shim(fused=bool(seed % 2)):
        from pde.backends import get_backend

        import pde_hip.pypde_plugin  # noqa: F401

        hip = get_backend("hip")
        rate = hip.native_to_numpy(eq.make_pde_rhs(state, backend="hip")(hip.numpy_to_native(state.data), 0.0))
    assert max_rel(rate, expect) < 1e-10, rhs
