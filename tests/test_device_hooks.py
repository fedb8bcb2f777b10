"""Post-step hooks traced onto the device (``pde_hip/hooks.py``; VERDICT r3 "missing #4"; reference: hooks compiled into the jitted
loops, pde/backends/numba/_solvers.py:22-64).  The tracer against numpy directly, and - through the real py-pde on the host shim - runs
with traced hooks against the reference's numpy backend; hooks that cannot be traced keep the host round trip."""

from __future__ import annotations

import sys

import numpy as np
import pytest
from refpath import REF  # noqa: E402

from pde_hip.hooks import trace_hook


def _numpy_of(expr: str):
    import sympy as sp

    return sp.lambdify([sp.Symbol("c"), sp.Symbol("t")], sp.sympify(expr), modules="numpy")


def _check(hook, t=0.7):
    rng = np.random.default_rng(0)
    data = rng.uniform(-1.5, 1.5, (6, 7))
    expr = trace_hook(hook, None, data.shape, data.dtype)
    assert expr is not None
    host = data.copy()
    res = hook(host, t, None)
    if res is not None:
        host = res[0] if isinstance(res, tuple) else res
    np.testing.assert_allclose(np.broadcast_to(_numpy_of(expr)(data, t), data.shape), host, rtol=1e-15, atol=0)


def test_tracer_reproduces_typical_hooks():
    def clip_in_place(state_data, t, data):
        state_data[state_data < 0] = 0
        state_data[state_data > 1] = 1

    def clip_call(state_data, t, data):
        np.clip(state_data, -0.5, 0.8, out=state_data)
        return state_data, data

    def relax(state_data, t, data):
        mask = (state_data > 0.5) | (state_data < -0.5)
        state_data[mask] *= 0.5
        state_data += 0.01 * t
        return state_data, None

    def where(state_data, t, data):
        return np.where(np.abs(state_data) > 1, np.sign(state_data), state_data) * (1 + 0.1 * np.tanh(t)), None

    def bounded(state_data, t, data):
        return np.minimum(np.maximum(state_data, -1.0), 1.0 + 0.2 * t)

    def through_a_view(state_data, t, data):
        view = state_data[:]          # a VIEW in numpy: changes reach the state (the tracer handed out a copy)
        view += 0.25
        other = state_data[...]
        other[other > 1] = 1

    for hook in (clip_in_place, clip_call, relax, where, bounded, through_a_view):
        _check(hook)


def test_tracer_refuses_what_is_not_pointwise():
    def reduction(state_data, t, data):
        state_data -= state_data.mean()

    def control_flow(state_data, t, data):
        if (state_data > 1).any():
            raise StopIteration

    def counts(state_data, t, data):
        i = state_data > 0.8
        data += (state_data[i] - 0.8).sum()
        state_data[i] = 0.8
        return state_data, data

    def single_cell(state_data, t, data):
        state_data[0, 0] = 1.0

    def time_branch(state_data, t, data):
        if t > 1:
            state_data *= 2

    for hook in (reduction, control_flow, counts, single_cell, time_branch):
        assert trace_hook(hook, 0.0, (4, 4), np.float64) is None


if not (REF / "pde").exists():
    pytest.skip("py-pde (reference) not available", allow_module_level=True)
if str(REF) not in sys.path:
    sys.path.append(str(REF))

import pde  # noqa: E402
import shimlib  # noqa: E402
from helpers import max_rel  # noqa: E402


@pytest.mark.parametrize("solver,adaptive", [("euler", False), ("runge-kutta", False), ("runge-kutta", True), ("euler", True)])
def test_traced_hook_runs_on_the_device(monkeypatch, solver, adaptive):
    """`pde.PDE(..., post_step_hook=...)` with a clipping hook that also depends on time: no download / upload per step (counted), equal
    to the reference's numpy backend (its solver with scipy operators) incl. step counts."""
    monkeypatch.setitem(pde.config, "default_backend", "scipy")

    def post_step_hook(state_data, t):
        state_data[state_data > 0.8] = 0.8
        state_data[state_data < 0.1 * np.tanh(t)] = 0.1 * np.tanh(t)
        return state_data

    grid = pde.UnitGrid([12, 10], periodic=[True, False])
    state = pde.ScalarField.random_uniform(grid, 0.0, 1.0, rng=np.random.default_rng(5))
    bc = {"x": "periodic", "y": {"derivative": 0.1}}

    class Restated(pde.PDEBase):      # the equation for the numpy backend (pde.PDE takes its operators from numba there)
        def evolution_rate(self, state, t=0):
            return 0.6 * state.laplace(bc) - 0.1 * state

        def make_post_step_hook(self, state, backend="numpy"):
            def hook(state_data, t, data):
                return post_step_hook(state_data, t), None

            return hook, None

    kw = dict(t_range=0.3, dt=0.01, solver=solver, adaptive=adaptive, tracker=None, ret_info=True)
    ref, iref = Restated().solve(state, backend="numpy", **kw)
    with shimlib.use_shim():
        import pde_hip.pypde_plugin  # noqa: F401
        from pde_hip import device

        counts = {"down": 0}
        orig = device.DeviceArray.get_valid

        def counting(self, *a, **k):
            counts["down"] += 1
            return orig(self, *a, **k)

        monkeypatch.setattr(device.DeviceArray, "get_valid", counting)
        eq = pde.PDE({"c": "0.6 * laplace(c) - 0.1 * c"}, bc=bc, post_step_hook=post_step_hook)
        res, info = eq.solve(state, backend="hip", **kw)
        got = np.array(res.data)
    assert info["solver"]["steps"] == iref["solver"]["steps"] > 3
    assert max_rel(got, ref.data) < 1e-10 and got.max() <= 0.8
    assert counts["down"] <= 2, counts      # the final read(s) only - with the host path: one per step


@pytest.mark.parametrize("solver,adaptive", [("euler", False), ("runge-kutta", False), ("runge-kutta", True), ("euler", True)])
@pytest.mark.parametrize("interval", [0.02, 0.05])
def test_traced_hook_with_a_tracker_that_splits_the_run(monkeypatch, solver, adaptive, interval):
    """with a tracker the stepper is called once per interval of 2-5 steps; the hook of round 4 recycled the array the
    state had left in the PREVIOUS call - which can be the caller's `state_data`, the next call's output buffer (`cur is nxt`: an
    in-place stencil sweep, 1e-2 off for adaptive RK).  Every stored frame equals the reference's numpy run, and no right-hand side
    is ever applied to its own output array."""
    monkeypatch.setitem(pde.config, "default_backend", "scipy")

    def post_step_hook(state_data, t):
        state_data[state_data > 0.8] = 0.8
        state_data[state_data < 0.05] = 0.05
        return state_data

    grid = pde.UnitGrid([12, 10], periodic=[True, False])
    state = pde.ScalarField.random_uniform(grid, 0.0, 1.0, rng=np.random.default_rng(11))
    bc = {"x": "periodic", "y": {"derivative": 0.1}}

    class Restated(pde.PDEBase):
        def evolution_rate(self, state, t=0):
            return 0.6 * state.laplace(bc) - 0.1 * state

        def make_post_step_hook(self, state, backend="numpy"):
            def hook(state_data, t, data):
                return post_step_hook(state_data, t), None

            return hook, None

    kw = dict(t_range=0.3, dt=0.01, solver=solver, adaptive=adaptive, ret_info=True)
    sref = pde.MemoryStorage()
    ref, iref = Restated().solve(state, backend="numpy", tracker=sref.tracker(interval), **kw)
    with shimlib.use_shim():
        import pde_hip.pypde_plugin  # noqa: F401
        from pde_hip import expr

        aliased = []
        orig = expr.ExpressionRhs.apply

        def watching(self, state_, out, wrap="rate", *a, **k):
            if wrap != "rate" and state_.ptr == out.ptr:      # (the pointwise hook pass itself runs in place, wrap "rate")
                aliased.append(wrap)
            return orig(self, state_, out, wrap, *a, **k)

        monkeypatch.setattr(expr.ExpressionRhs, "apply", watching)
        shim_store = pde.MemoryStorage()
        eq = pde.PDE({"c": "0.6 * laplace(c) - 0.1 * c"}, bc=bc, post_step_hook=post_step_hook)
        res, info = eq.solve(state, backend="hip", tracker=shim_store.tracker(interval), **kw)
        got = np.array(res.data)
        frames = [np.array(f) for f in shim_store.data]
    assert not aliased, aliased
    assert info["solver"]["steps"] == iref["solver"]["steps"] > 3
    assert len(frames) == len(sref.data) >= 6
    for a, b in zip(frames, sref.data):
        assert max_rel(a, b) < 1e-10
    assert max_rel(got, ref.data) < 1e-10 and got.max() <= 0.8


def test_untraceable_hook_keeps_the_host_path(monkeypatch):
    monkeypatch.setitem(pde.config, "default_backend", "scipy")
    calls = []

    def post_step_hook(state_data, t):
        calls.append(float(state_data.max()))     # a reduction: host
        state_data[state_data > 0.8] = 0.8
        return state_data

    grid = pde.UnitGrid([8, 8], periodic=True)
    state = pde.ScalarField.random_uniform(grid, 0.5, 1.0, rng=np.random.default_rng(6))
    with shimlib.use_shim():
        import pde_hip.pypde_plugin  # noqa: F401

        res = pde.PDE({"c": "laplace(c)"}, post_step_hook=post_step_hook).solve(state, t_range=0.05, dt=0.01, backend="hip", tracker=None)
        assert np.array(res.data).max() <= 0.8
    assert len(calls) >= 5
