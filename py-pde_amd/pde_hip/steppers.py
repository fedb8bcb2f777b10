"""Time steppers of the backend: Euler / RK4 / RKF45 / adaptive Euler / Adams-Bashforth / Euler-Maruyama steppers around the C loops,
post-step hooks, ``make_inner_stepper`` / ``make_stepper`` - :class:`StepperMixin`.  Split from ``backend.py`` in round 6 (no behaviour change).

Reference: ``pde/solvers/euler.py:66-283``, ``pde/solvers/runge_kutta.py:29-156``, ``pde/backends/numba/_solvers.py:22-466``, ``pde/backends/base.py:728-755``.
"""

from __future__ import annotations

import ctypes as C
import inspect
import logging
import os
from collections import defaultdict
from typing import Any, Callable, NamedTuple

import numpy as np

from . import _abi
from ._lib import require_device
from .device import DeviceArray, DeviceBuffer, DeviceScalar, GridInfo, ptr_array
from .faces import real_dtype_of
from .resident import ResidentState, _config_get
from .rhs import SpecRhs

_logger = logging.getLogger("pde_hip.backend")


class StepperMixin:
    """The stepper-facing methods of :class:`~pde_hip.backend.HipBackendMixin`."""

    def _make_expression_stepper(self, solver, state, erhs=None, post_step=None, reduce_error=None, scheme=None):
        """Python-level twin of the C steppers for expression right-hand sides: the same update rules
        (pde/solvers/euler.py:172-175, runge_kutta.py:52-61, :135-153) with the RHS evaluated by the
        run-time specialised kernels; the Euler update / RK stage scaling is folded into the last pass.
        ``erhs``: any evaluator with the interface of :class:`~pde_hip.expr.ExpressionRhs` (default: the expression
        of ``solver.pde``; :class:`SpecRhs` for the class PDEs when their BCs depend on time).
        ``post_step(array, t) -> array``: the PDE's post-step hook (after every fixed step with the time the step started at,
        ``pde/solvers/base.py:266-272``; after every accepted adaptive step with the new time,
        ``pde/backends/numba/_solvers.py:262-270``); with a hook every step is a single sweep."""
        from .solvers import OnlineStatistics, make_dt_adjuster

        if erhs is None:
            erhs = self.make_expression_rhs(solver.pde, state)
        info, lib, stream = erhs.info, self._lib, self.stream
        ncomp = int(getattr(erhs, "ncomp", 1))                      # > 1: multi-field PDE (SystemRhs)
        comp_shape = (ncomp,) if ncomp > 1 else ()
        # (`scheme`: "euler" / "runge-kutta" for callers without one of the solver classes, e.g. the decomposed steppers)
        is_rk = (scheme == "runge-kutta") if scheme is not None else solver.__class__.__name__ == "RungeKuttaSolver"
        adaptive = bool(getattr(solver, "adaptive", False))
        nwork = (7 if adaptive else 5) if is_rk else (3 if adaptive else 1)   # adaptive Euler: rate, half step, slope scratch
        # complex states (planar (re, im) pairs, SystemRhs.complex_pairs): arrays that may hold the state hand out complex host data
        # (hooks); the error norm of the adaptive schemes is the modulus `np.abs(complex)` - taken from an explicit error field
        is_complex = bool(getattr(erhs, "complex_pairs", False))
        if is_complex:
            comp_shape = (ncomp // 2, 2)
            nwork += 1 if adaptive else 0      # the error field
        work = [DeviceArray(info, comp_shape, complex_pairs=is_complex) for _ in range(nwork)]
        B = [[1 / 4], [3 / 32, 9 / 32], [1932 / 2197, -7200 / 2197, 7296 / 2197], [439 / 216, -8.0, 3680 / 513, -845 / 4104],
             [-8 / 27, 2.0, -3544 / 2565, 1859 / 4104, -11 / 40]]
        A = [0.0, 1 / 4, 3 / 8, 12 / 13, 1.0, 1 / 2]

        def lincomb(out, y, coefs, ks):
            cf = (C.c_double * len(coefs))(*coefs)
            lib.lincomb(info.ref, ncomp, out.ptr, y.ptr, len(ks), cf, ptr_array(ks), stream)

        def rk4_step(y, t, dt):
            # every stage in one sweep where the kernels cover it (slope + the combination that follows, like
            # pdehip_rk4_step): the array of k4 serves as the second stage input, k4 itself stays in registers
            k1, k2, k3, k4, tmp = work[:5]
            if not erhs.apply_stage(y, k1, dt, t, 0, y, [], [], 0.5, tmp):
                lincomb(tmp, y, [0.5], [k1])
            if not erhs.apply_stage(tmp, k2, dt, t + 0.5 * dt, 0, y, [], [], 0.5, k4):
                lincomb(k4, y, [0.5], [k2])
            if not erhs.apply_stage(k4, k3, dt, t + 0.5 * dt, 0, y, [], [], 1.0, tmp):
                lincomb(tmp, y, [1.0], [k3])
            if not erhs.apply_stage(tmp, k4, dt, t + dt, 1, y, [k1, k2, k3], [], 0.0, y):
                lib.rk4_combine(info.ref, ncomp, y.ptr, k1.ptr, k2.ptr, k3.ptr, k4.ptr, stream)

        if not adaptive:
            dt = float(solver.info["dt"])
            cells = int(np.prod(info.shape))
            can_two = ncomp == 1 and getattr(erhs, "_two_ok", False) is not False

            def use_loop(steps: int) -> bool:
                # large grids whose expression runs two steps per sweep keep that (Python overhead is noise there)
                if is_rk or post_step is not None or not hasattr(erhs, "euler_loop") or os.environ.get("PDEHIP_EXPR_LOOP") == "0":
                    return False
                return not (can_two and cells > (1 << 21))

            def fixed_stepper(state_data: DeviceArray, t_start: float, t_end: float):
                steps = max(1, round((t_end - t_start) / dt))
                cur, nxt = state_data, work[0]
                i = 0
                try:
                    if use_loop(steps):
                        # the whole loop in ONE C call (captured as a hipGraph for long runs): a Python iteration per step
                        # costs 40-85 us where the kernels of a small grid need 2-5 us
                        done = erhs.euler_loop(cur, nxt, dt, t_start, steps)
                        if done is not None:
                            if done is not cur:
                                cur, nxt = nxt, cur
                            i = steps
                    if is_rk and post_step is None and hasattr(erhs, "rk_run") and os.environ.get("PDEHIP_EXPR_LOOP") != "0":
                        # the whole fixed-step RK4 loop in ONE C call (pdehip_jit_rk_run; reference: the jitted loop
                        # pde/backends/numba/_solvers.py:93-118 around pde/solvers/runge_kutta.py:29-66)
                        if erhs.rk_run(cur, None, work[:5], None, dt, t_start, steps) is not None:
                            i = steps
                    while i < steps:
                        t = t_start + i * dt
                        if is_rk:
                            rk4_step(cur, t, dt)
                        elif post_step is None and i + 2 <= steps and erhs.euler2(cur, nxt, dt):   # two steps per sweep (one-pass expressions)
                            cur, nxt = nxt, cur
                            i += 1
                        else:
                            erhs.apply(cur, nxt, "euler", dt, t)
                            cur, nxt = nxt, cur
                        i += 1
                        if post_step is not None:
                            cur = post_step(cur, t, nxt) if getattr(post_step, "wants_prev", False) else post_step(cur, t)
                finally:
                    # also when a hook ends the run with StopIteration: the caller's array holds the latest state
                    if cur is not state_data:
                        lib.memcpy_d2d(state_data.ptr, cur.ptr, state_data.nbytes, stream)
                    solver.info["steps"] += i
                return state_data, t_start + (steps - 1) * dt + dt

            return fixed_stepper

        solver.info["dt_adaptive"] = True
        solver.info.setdefault("dt_statistics", OnlineStatistics())
        adjust_dt = make_dt_adjuster(solver.dt_min, solver.dt_max)
        tolerance, dt_min = float(solver.tolerance), float(solver.dt_min)
        err_dev, ynew0 = DeviceScalar(), DeviceArray(info, comp_shape, complex_pairs=is_complex)

        def attempt_complex(y, ynew, t, dt_step) -> float:
            """The attempts below for complex states: new state and error FIELD with the pointwise kernels, then max |error| as the
            modulus over the (re, im) pairs (pdehip_max_abs_pairs) - `np.abs(...).max()` of a complex array in the reference."""
            efield = work[-1]
            if is_rk:
                ks, tmp = work[:6], work[6]
                src = y
                for s_, b in enumerate(B):
                    erhs.apply(src, ks[s_], "scaled", dt_step, t + A[s_] * dt_step)
                    lincomb(tmp, y, b, ks[: s_ + 1])
                    src = tmp
                erhs.apply(src, ks[5], "scaled", dt_step, t + A[5] * dt_step)
                lincomb(ynew, y, [25 / 216, 1408 / 2565, 2197 / 4104, -1 / 5], [ks[0], ks[2], ks[3], ks[4]])          # runge_kutta.py:150
                cf = (C.c_double * 5)(1 / 360, -128 / 4275, -2197 / 75240, 1 / 50, 2 / 55)                                # runge_kutta.py:147
                lib.lincomb(info.ref, ncomp, efield.ptr, None, 5, cf, ptr_array([ks[0], ks[2], ks[3], ks[4], ks[5]]), stream)
            else:
                rate, half, kmid = work[0], work[1], work[2]
                h = 0.5 * dt_step
                erhs.apply(half, kmid, "scaled", h, t + h)
                lincomb(ynew, half, [1.0], [kmid])              # step_small += 0.5 * dt * rate_midpoint
                lincomb(efield, y, [dt_step], [rate])            # step_large
                lincomb(efield, efield, [-1.0], [ynew])          # step_large - step_small
            lib.max_abs_pairs(info.ref, ncomp // 2, efield.ptr, err_dev.ptr, stream)
            if reduce_error is not None:
                reduce_error(err_dev)
            return err_dev.value(stream)

        def attempt(y, ynew, t, dt_step) -> float:
            if is_complex:
                return attempt_complex(y, ynew, t, dt_step)
            if is_rk:
                # stages 1-5: slope + next stage input in one sweep (inputs alternate between tmp and ynew, which is free
                # until the last sweep); stage 6: new state + error norm with k6 in registers (like pdehip_rkf45_attempt)
                ks, tmp = work[:6], work[6]
                src, dst = y, tmp
                for s_, b in enumerate(B):
                    if not erhs.apply_stage(src, ks[s_], dt_step, t + A[s_] * dt_step, 0, y, ks[:s_], b[:s_], b[s_], dst):
                        lincomb(dst, y, b, ks[: s_ + 1])
                    src, dst = dst, (ynew if dst is tmp else tmp)
                if not erhs.apply_stage(src, ks[5], dt_step, t + A[5] * dt_step, 2, y, [ks[0], ks[2], ks[3], ks[4]], [], 0.0, ynew, err_dev):
                    lib.rkf45_combine(info.ref, ncomp, y.ptr, ynew.ptr, ptr_array(ks), err_dev.ptr, stream)
            else:
                # second half of the reference's adaptive Euler attempt (pde/backends/numba/_solvers.py:385-394): `work[1]` holds
                # step_small = y + dt/2 * rate; the sweep adds dt/2 * rhs(step_small, t + dt/2) and takes the error norm against
                # step_large = y + dt * rate, which is never stored (stage kind 4)
                rate, half, kmid = work[0], work[1], work[2]
                h = 0.5 * dt_step
                if not erhs.apply_stage(half, kmid, h, t + h, 4, y, [rate, half], [dt_step, 0.0], 0.0, ynew, err_dev):
                    lib.euler_adaptive_combine(info.ref, ncomp, y.ptr, rate.ptr, dt_step, half.ptr, kmid.ptr, ynew.ptr, err_dev.ptr, stream)
            if reduce_error is not None:
                reduce_error(err_dev)     # MAX over the ranks of a decomposed run, on the device, NaN wins (pde/backends/base.py:678-712)
            return err_dev.value(stream)

        ctl = None
        # (decomposed grids: the C loops reduce the error over the ranks themselves when the passes carry their exchange descriptor)
        reduces_in_c = reduce_error is None or bool(getattr(erhs, "reduces_error_in_loops", False))
        if post_step is None and hasattr(erhs, "rk_run") and reduces_in_c and os.environ.get("PDEHIP_EXPR_LOOP") != "0":
            # the adaptive loop itself in C (pdehip_jit_rk_run: pde/backends/numba/_solvers.py:199-319 is jitted in the reference)
            from .solvers import AdaptiveStatistics

            ctl = _abi.Adaptive()
            ctl.tolerance, ctl.dt_min, ctl.dt_max = tolerance, dt_min, float(solver.dt_max)

        def adaptive_stepper(state_data: DeviceArray, t_start: float, t_end: float):
            nonlocal ctl
            if ctl is not None:
                ctl.t_start, ctl.t_end, ctl.dt = float(t_start), float(t_end), float(solver.info["dt"])
                before = int(ctl.steps)
                try:
                    if is_rk:
                        # (complex states: one more array, the error field of the modulus norm - round 5)
                        res = erhs.rk_run(state_data, ynew0, work[:7] + ([work[-1]] if is_complex else []), err_dev, 0.0, 0.0, 0, ctl)
                    else:   # the reference's adaptive Euler loop in one C call (pdehip_jit_euler_adaptive_run)
                        res = erhs.rk_run(state_data, ynew0, work[:3] + ([work[-1]] if is_complex else []), err_dev, 0.0, 0.0, 0, ctl, euler_adaptive=True)
                finally:
                    solver.info["steps"] += int(ctl.steps) - before
                    solver.info["attempts"] = int(ctl.attempts)
                if res is not None:
                    if res is not state_data:
                        lib.memcpy_d2d(state_data.ptr, res.ptr, state_data.nbytes, stream)
                    solver.info["dt"] = float(ctl.dt)
                    solver.info["dt_statistics"] = AdaptiveStatistics(ctl)
                    return state_data, float(ctl.t_last)
                ctl = None      # not available for this right-hand side (integrals, function-valued conditions): Python loop
            dt_opt = float(solver.info["dt"])
            t, steps = t_start, 0
            stats = solver.info["dt_statistics"]
            cur, nxt = state_data, ynew0   # an accepted attempt swaps the roles (no copy of the field per step)
            # Adaptive Euler is the reference's own loop (pde/backends/numba/_solvers.py:374-433, pde/solvers/euler.py:222-280; C twin
            # csrc/pdehip_rk_loops.h `euler_adaptive_run`): the rate of the current state is carried from attempt to attempt and,
            # after an accepted attempt, evaluated at the time BEFORE `t += dt` - here lazily at the start of the next attempt, in
            # the sweep that also writes the first half step; with a hook eagerly, before the hook sees (and may change) the state.
            have_rate, t_rate = False, t_start
            try:
                while True:
                    dt_step = max(min(dt_opt, t_end - t), dt_min)
                    if not is_rk:
                        rate, half = work[0], work[1]
                        h = 0.5 * dt_step
                        if is_complex and not have_rate:
                            erhs.apply(cur, rate, "rate", 0.0, t_rate)
                            have_rate = True
                        if have_rate or not erhs.apply_stage(cur, rate, 1.0, t_rate, 0, cur, [], [], h, half):
                            lincomb(half, cur, [h], [rate])
                        have_rate = True
                    error_rel = attempt(cur, nxt, t, dt_step) / tolerance
                    if error_rel <= 1:
                        steps += 1
                        t_rate = t
                        t += dt_step
                        cur, nxt = nxt, cur
                        have_rate = False
                        if post_step is not None:
                            if not is_rk:
                                erhs.apply(cur, work[0], "rate", 0.0, t_rate)   # `rate = rhs_pde(step_small, t)` precedes the hook (:402-411)
                                have_rate = True
                            cur = post_step(cur, t)
                        stats.add(dt_step)
                    if t < t_end:
                        dt_opt = adjust_dt(dt_step, error_rel)
                    else:
                        break
            finally:
                if cur is not state_data:
                    lib.memcpy_d2d(state_data.ptr, cur.ptr, state_data.nbytes, stream)
                solver.info["dt"] = dt_opt
                solver.info["steps"] += steps
            return state_data, t

        return adaptive_stepper

    # --- steppers ----------------------------------------------------------------------------------------------
    def _make_adams_bashforth_stepper(self, solver, spec):
        """Two-step Adams-Bashforth (pde/solvers/adams_bashforth.py:31-70, pde/backends/numba/_solvers.py:121-196).

        The reference re-evaluates ``rhs(state_prev)`` in every step; it equals the ``rhs_cur`` of the step before
        bit for bit, so it is kept instead: one right-hand side per step.  Rates are ``pdehip_rhs_scaled`` with dt = 1.
        """
        info, lib, stream = spec.info, self._lib, self.stream
        dt = float(solver.info["dt"])
        rates = [DeviceArray(info), DeviceArray(info)]   # [current, previous], roles swap every step
        tmp = DeviceArray(info)
        minus_dt = (C.c_double * 1)(-dt)
        first, one_sweep = [True], [True]

        def fixed_stepper(state_data: DeviceArray, t_start: float, t_end: float):
            steps = max(1, round((t_end - t_start) / dt))
            if first[0]:
                # state_prev = state - dt * rhs(state)  ->  rate_prev = rhs(state_prev)
                spec.c.t = float(t_start)              # every rate at its own time (adams_bashforth.py:45-46, :64): t, then t - dt
                lib.rhs_scaled(info.ref, spec.ref, state_data.ptr, rates[0].ptr, 1.0, stream)
                lib.lincomb(info.ref, 1, tmp.ptr, state_data.ptr, 1, minus_dt, ptr_array([rates[0]]), stream)
                spec.c.t = float(t_start) - dt
                lib.rhs_scaled(info.ref, spec.ref, tmp.ptr, rates[1].ptr, 1.0, stream)
                first[0] = False
            cur, nxt = state_data, tmp
            fused = C.c_int(0)
            for i in range(steps):
                spec.c.t = t_start + i * dt
                # rate and update in one sweep where the kernels cover it (state ping-pongs), else two kernels in place
                if one_sweep[0]:
                    lib.ab2_step(info.ref, spec.ref, cur.ptr, nxt.ptr, rates[0].ptr, rates[1].ptr, dt, C.byref(fused), stream)
                    one_sweep[0] = bool(fused.value)
                if one_sweep[0]:
                    cur, nxt = nxt, cur
                else:
                    lib.rhs_scaled(info.ref, spec.ref, cur.ptr, rates[0].ptr, 1.0, stream)
                    lib.ab2_combine(info.ref, 1, cur.ptr, rates[0].ptr, rates[1].ptr, dt, stream)
                rates.reverse()
            if cur is not state_data:
                lib.memcpy_d2d(state_data.ptr, cur.ptr, state_data.nbytes, stream)
            solver.info["steps"] += steps
            return state_data, t_start + (steps - 1) * dt + dt

        return fixed_stepper

    def _make_adams_bashforth_expression_stepper(self, solver, erhs):
        """Two-step Adams-Bashforth (pde/solvers/adams_bashforth.py:31-70, pde/backends/numba/_solvers.py:121-196) around any evaluator
        with the interface of :class:`~pde_hip.expr.ExpressionRhs` (expression PDEs, systems, complex states as real systems).  Like
        the class version above, ``rhs(state_prev, t - dt)`` is the rate of the step before, kept instead of being evaluated again."""
        info, lib, stream = erhs.info, self._lib, self.stream
        ncomp = int(getattr(erhs, "ncomp", 1))
        is_complex = bool(getattr(erhs, "complex_pairs", False))
        comp_shape = ((ncomp // 2, 2) if is_complex else (ncomp,)) if ncomp > 1 else ()
        dt = float(solver.info["dt"])
        rates = [DeviceArray(info, comp_shape, complex_pairs=is_complex) for _ in range(2)]   # [current, previous], roles swap every step
        tmp = DeviceArray(info, comp_shape, complex_pairs=is_complex)
        minus_dt = (C.c_double * 1)(-dt)
        first = [True]

        def fixed_stepper(state_data: DeviceArray, t_start: float, t_end: float):
            steps = max(1, round((t_end - t_start) / dt))
            if first[0]:
                # state_prev = state - dt * rhs(state, t)  ->  rate_prev = rhs(state_prev, t - dt)   (adams_bashforth.py:62-66)
                erhs.apply(state_data, rates[0], "rate", 0.0, float(t_start))
                lib.lincomb(info.ref, ncomp, tmp.ptr, state_data.ptr, 1, minus_dt, ptr_array([rates[0]]), stream)
                erhs.apply(tmp, rates[1], "rate", 0.0, float(t_start) - dt)
                first[0] = False
            for i in range(steps):
                erhs.apply(state_data, rates[0], "rate", 0.0, t_start + i * dt)
                lib.ab2_combine(info.ref, ncomp, state_data.ptr, rates[0].ptr, rates[1].ptr, dt, stream)
                rates.reverse()
            solver.info["steps"] += steps
            return state_data, t_start + (steps - 1) * dt + dt

        return fixed_stepper

    def make_gaussian_noise(self, field, *, rng=None):
        """``noise() -> DeviceArray`` of independent standard-normal values with the shape of ``field.data``
        (``BackendBase.make_gaussian_noise``, pde/backends/base.py:714-726; numba: pde/backends/numba/backend.py, torch:
        pde/backends/torch/backend.py:603-625).  Device generator of ``pdehip_add_gaussian_noise`` (Philox4x32-10 +
        Box-Muller) seeded from ``rng`` like the torch backend; every call advances the counter."""
        grid = field.grid
        info = self.grid_info(grid, field.dtype)
        nd = grid.num_axes
        comp_shape = tuple(field.data.shape[: field.data.ndim - nd])
        seed = int((rng if rng is not None else np.random.default_rng()).integers(0, 2**32))
        counter = [0]
        lib = self._lib

        def noise() -> DeviceArray:
            out = DeviceArray(info, comp_shape)   # zero-initialised
            lib.add_gaussian_noise(info.ref, out.ncomp, out.ptr, 1.0, seed, counter[0], 0, self.stream)
            counter[0] += 1
            return out

        return noise

    def _make_noise_step(self, solver, state):
        """Noise increment of an Euler-Maruyama step as ``add_noise(array: DeviceArray)``, or None for deterministic equations.

        Covers the reference's standard case — additive Gaussian white noise of constant variance ``eq.noise``
        (``SDEBase.make_noise_variance``, ``pde/pdes/base.py:634-722``) in ``EulerSolver`` with a fixed step
        (``pde/solvers/euler.py:66-147``): ``state += sqrt(dt) * sqrt(noise / cell_volume) * dW``; additive noise has no drift
        correction in any interpretation.  dW comes from the device generator of ``pdehip_add_gaussian_noise`` seeded from
        ``eq.rng`` (like the torch backend, ``pde/backends/torch/backend.py:603-625``); realisations are therefore not those
        of the numba backend, only their statistics agree.  Everything else (state-dependent variance, noise realisations,
        Milstein, adaptive steps) raises like the reference / ``NotImplementedError``."""
        eq = solver.pde
        if not getattr(eq, "is_sde", False):
            return None
        solver_name = solver.__class__.__name__
        if bool(getattr(solver, "adaptive", False)):
            msg = "Cannot use adaptive stepping with stochastic equation"   # pde/solvers/base.py:446-449
            raise RuntimeError(msg)
        if solver_name not in {"EulerSolver", "ExplicitSolver", "MilsteinSolver"}:
            msg = f"Backend `{self.name}` does not support stochastic equations with {solver_name}"
            raise NotImplementedError(msg)
        custom_variance = False
        for cls in type(eq).__mro__:
            if "make_noise_variance" in vars(cls):
                custom_variance = cls.__name__ not in {"SDEBase", "PDEBase"}
                break
        if getattr(eq, "use_noise_realization", False):
            # Noise given as a REALISATION (pde/pdes/base.py:578, pde/solvers/euler.py:99-127: `state += sqrt(dt) * realization(state_old, t)`):
            # arbitrary Python on host arrays - the reference's own device backend refuses it (pde/backends/torch/_solvers.py:312-314).  Here:
            # a host round trip per step (the old state down, the realisation up), warned like the hooks that cannot be traced.
            realization = eq.make_noise_realization(state, backend=self)
            _logger.warning("noise realisations of %s are user code on host arrays: the state crosses PCIe twice per step", type(eq).__name__)
            dt_sqrt = (C.c_double * 1)(float(np.sqrt(float(solver.info["dt"]))))
            has_var = not np.allclose(np.asarray(getattr(eq, "noise", 0), dtype=float), 0, atol=1e-14)
            if getattr(eq, "use_noise_variance", True) and has_var:
                msg = f"Backend `{self.name}`: a noise variance next to a noise realisation is not supported"
                raise NotImplementedError(msg)
            ninfo = self.grid_info(state.grid, state.dtype)
            comp = tuple(np.shape(state.data))[: np.ndim(state.data) - len(ninfo.shape)]

            def add_realization(arr: DeviceArray, prev=None, t: float = 0.0) -> None:
                host_old = (prev if prev is not None else arr).get_valid(stream=self.stream)
                noise = realization(host_old, t)
                if noise is None:
                    return
                up = DeviceArray(ninfo, comp).set_valid(np.ascontiguousarray(np.broadcast_to(noise, host_old.shape), dtype=ninfo.dtype), self.stream)
                self._lib.lincomb(ninfo.ref, int(np.prod(comp)) if comp else 1, arr.ptr, arr.ptr, 1, dt_sqrt, ptr_array([up]), self.stream)

            solver.info["stochastic"] = True
            return add_realization
        if not getattr(eq, "use_noise_variance", True):
            msg = f"Backend `{self.name}`: a stochastic equation without noise variance and without noise realisation"
            raise NotImplementedError(msg)
        if custom_variance:
            return self._make_traced_noise_step(solver, state)
        grid = state.grid
        nd = grid.num_axes
        ncomp = int(np.prod(state.data.shape[: state.data.ndim - nd])) if state.data.ndim > nd else 1
        try:
            # one variance for all fields or one per field of a collection (pde/pdes/pde.py:266-281, base.py:634-722)
            noise = np.broadcast_to(np.asarray(getattr(eq, "noise", 0), dtype=float), (ncomp,))
        except ValueError:
            noise = None
        if noise is None or (noise < 0).any():
            msg = f"Backend `{self.name}` needs one non-negative noise variance per field"
            raise NotImplementedError(msg)
        info = self.grid_info(grid, state.dtype)
        cell_volume = float(np.prod(grid.discretization))
        cells = int(np.prod(grid.shape))
        dt = float(solver.info["dt"])
        scales = [float(np.sqrt(dt) * np.sqrt(v / cell_volume)) for v in noise]
        rng = getattr(eq, "rng", None)
        seed = int((rng if rng is not None else np.random.default_rng()).integers(0, 2**32))   # like the torch backend (torch/backend.py:619)
        counter = [0]
        lib = self._lib

        def add_noise(arr: DeviceArray, prev=None, t: float = 0.0) -> None:
            if ncomp == 1:
                lib.add_gaussian_noise(info.ref, 1, arr.ptr, scales[0], seed, counter[0], 0, self.stream)
            else:
                # every field its own variance; the cell offset keeps the fields' random streams apart
                for k in range(ncomp):
                    if scales[k] != 0:
                        lib.add_gaussian_noise(info.ref, 1, arr.flat().component(k).ptr, scales[k], seed, counter[0], k * cells, self.stream)
            counter[0] += 1

        solver.info["stochastic"] = True
        return add_noise

    def _make_traced_noise_step(self, solver, state):
        """Euler-Maruyama increment for a noise variance that depends on the field (``make_noise_variance`` overridden by the user,
        ``pde/pdes/base.py:634-722``; multiplicative noise): ``add_noise(new, old, t)``.

        The user's function ``noise_variance(state_data, t)`` is Python; like ``user_funcs`` it is TRACED once with a symbolic field
        and compiled into one pointwise kernel that applies the reference's update (``pde/solvers/euler.py:112-141``) to the
        deterministic step: ``new += sqrt(dt) * sqrt(variance(old, t) / cell_volume) * dW`` and, for interpretations other than
        Ito, ``+ 0.5 * dt * alpha * d variance / d field (old, t) / cell_volume``.  The variance is evaluated on the state BEFORE
        the step, like the reference does.  dW comes from the device generator (see :meth:`_make_noise_step`)."""
        import sympy as sp

        from .expr import ExpressionPlan, ExpressionRhs

        eq = solver.pde
        grid = state.grid
        if state.__class__.__name__ != "ScalarField":
            msg = f"Backend `{self.name}`: a noise variance that depends on the field is supported for scalar fields"
            raise NotImplementedError(msg)
        alpha = float(getattr(eq, "_noise_drift_factor", 0.0))
        milstein = solver.__class__.__name__ == "MilsteinSolver"     # pde/solvers/milstein.py:103-127: always with the derivative
        need_diff = alpha != 0 or milstein
        c, t = sp.Symbol("pdehip_c", real=True), sp.Symbol("t", real=True)
        try:
            try:
                func = eq.make_noise_variance(state, backend=self, ret_diff=need_diff)
            except TypeError:
                func = eq.make_noise_variance(state, backend=self)
            traced = func(c, t)
            var, dvar = (traced if need_diff else (traced, 0))
            var, dvar = sp.sympify(var), sp.sympify(dvar)
        except NotImplementedError:
            raise
        except Exception as err:   # noqa: BLE001 - whatever the user's code raises on symbolic input
            msg = (f"hip backend: the noise variance of {eq.__class__.__name__} cannot be traced symbolically ({type(err).__name__}: {err}); "
                   "it must work on sympy expressions (arithmetic, sympy functions)")
            raise NotImplementedError(msg) from err
        unknown = (var.free_symbols | dvar.free_symbols) - {c, t}
        if unknown:
            msg = f"hip backend: the noise variance of {eq.__class__.__name__} depends on {sorted(map(str, unknown))}"
            raise NotImplementedError(msg)
        info = self.grid_info(grid, state.dtype)
        cell_volume = float(np.prod(grid.discretization))
        dt = float(solver.info["dt"])
        # sqrt(dt) * sqrt(var / V) * dW, the operations of pde/solvers/euler.py:132-133 in their order
        text = f"pdehip_unew + {float(np.sqrt(dt))!r} * sqrt(({sp.sstr(var)}) * {1.0 / cell_volume!r}) * pdehip_dw"
        if alpha != 0:
            text += f" + {0.5 * dt * alpha!r} * ({sp.sstr(dvar)}) * {1.0 / cell_volume!r}"
        if milstein:
            # + 0.25 * dvar / V * (dW**2 - dt) with dW = sqrt(dt) * xi   (pde/solvers/milstein.py:119-125)
            text += f" + 0.25 * ({sp.sstr(dvar)}) * {1.0 / cell_volume!r} * (({float(np.sqrt(dt))!r} * pdehip_dw)**2 - {dt!r})"
        plan = ExpressionPlan(text, "pdehip_c", {}, axes=tuple(grid.axes), aux=("pdehip_unew", "pdehip_dw"))
        dw = DeviceArray(info)
        erhs = ExpressionRhs(self, plan, info, {}, {"pdehip_unew": dw, "pdehip_dw": dw})   # (`unew` is bound per step)
        rng = getattr(eq, "rng", None)
        seed = int((rng if rng is not None else np.random.default_rng()).integers(0, 2**32))
        counter = [0]
        lib = self._lib

        def add_noise(arr: DeviceArray, prev=None, t: float = 0.0) -> None:
            if prev is None:
                msg = "internal: a field-dependent noise variance needs the state before the step"
                raise RuntimeError(msg)
            lib.memset(dw.ptr, 0, dw.nbytes, self.stream)
            lib.add_gaussian_noise(info.ref, 1, dw.ptr, 1.0, seed, counter[0], 0, self.stream)
            counter[0] += 1
            erhs.aux["aux:pdehip_unew"] = arr
            erhs.apply(prev, arr, "rate", 0.0, float(t))     # pointwise, in place on the new state

        add_noise.keepalive = (erhs, dw)   # type: ignore[attr-defined]
        solver.info["stochastic"] = True
        return add_noise

    def _make_host_post_step(self, solver, state):
        """The PDE's post-step hook (``pde/solvers/base.py:191-232``, ``pde/pdes/base.py:160-208``) as
        ``post_step(array: DeviceArray, t) -> DeviceArray``, or None when the PDE defines none.

        Hooks are user code written against numpy arrays (``state_data[i] = 1``, ``raise StopIteration`` ...), so they run
        on the HOST: the valid data is downloaded, handed to the hook, and uploaded again after every step — a full PCIe
        round trip per step, logged once as a warning.  ``StopIteration`` propagates to the controller
        (``pde/solvers/controller.py:235-240``); ``solver.info["post_step_data"]`` is kept up to date."""
        make_hook = getattr(solver.pde, "make_post_step_hook", None)
        if make_hook is None or not getattr(solver, "_use_post_step_hook", True):
            solver.info.setdefault("post_step_data", None)
            return None
        try:
            try:
                hook, data = make_hook(state, backend="numpy")
            except TypeError:
                hook, data = make_hook(state)          # mirror classes without the `backend` argument
        except NotImplementedError:
            solver.info["post_step_data"] = None   # no hook defined: the normal case
            return None
        solver.info["post_step_data"] = data
        device_hook = self._make_device_post_step(solver, state, hook, data)
        if device_hook is not None:
            return device_hook
        _logger.warning("post-step hook of %s runs on the host: the state crosses PCIe twice per step", solver.pde.__class__.__name__)

        def post_step(arr: DeviceArray, t: float) -> DeviceArray:
            host = arr.get_valid(stream=self.stream)
            try:
                result = hook(host, t, solver.info["post_step_data"])
            except StopIteration:
                # a hook may have changed the state IN PLACE before it ended the run (the reference's arrays are the state
                # itself, tests/pdes/test_pde_class.py:546-566): what it left behind is the final state
                arr.set_valid(np.asarray(host, dtype=arr.dtype), self.stream)
                raise
            if result is not None:                      # hooks may work in place and return nothing (older signature)
                host, solver.info["post_step_data"] = result
            arr.set_valid(np.asarray(host, dtype=arr.dtype), self.stream)
            return arr

        return post_step

    def _make_device_post_step(self, solver, state, hook, data):
        """The hook as ONE run-time compiled pointwise pass on the device (``pde_hip/hooks.py``: the hook is traced once with a symbolic
        array - masked assignment, ``np.clip`` / ``np.where`` / ``np.minimum`` ..., arithmetic with ``t``), or None when it cannot be
        traced (reductions, control flow on values, hook data that changes, states that are not one real scalar field): then the host
        round trip below.  The reference compiles hooks into its jitted loops (``pde/backends/numba/_solvers.py:22-64``).
        The trace CALLS the hook once with a symbolic array.  By default only hooks given as ``PDE(..., post_step_hook=f)`` are traced - the
        form the reference hands to its backend's compiler (``pde/pdes/pde.py:691-706``: compiled code has no Python side effects); a
        class that overrides ``make_post_step_hook`` may count calls or collect data in Python and keeps the host path unless
        ``PDEHIP_DEVICE_HOOKS=1`` asks for the trace (``=0``: never)."""
        mode = os.environ.get("PDEHIP_DEVICE_HOOKS", "auto")
        if mode == "0" or state.__class__.__name__ != "ScalarField" or np.dtype(state.dtype).kind != "f":
            return None
        if mode != "1":
            eq = solver.pde
            plain = getattr(eq, "post_step_hook", None) is not None and not any(
                "make_post_step_hook" in vars(c) for c in type(eq).__mro__ if c.__name__ not in ("PDE", "PDEBase", "object") and c.__module__ != "pde.pdes.pde")
            if not plain:
                return None
        from .expr import ExpressionPlan, ExpressionRhs
        from .hooks import trace_hook

        expr = trace_hook(hook, data, tuple(state.grid.shape), state.dtype)
        if expr is None:
            return None
        try:
            plan = ExpressionPlan(expr, "c", {}, axes=tuple(state.grid.axes))
            if plan.operators_used or plan.aux_used or len(plan.passes) != 1:
                return None
            erhs = ExpressionRhs(self, plan, self.grid_info(state.grid, state.dtype), {}, {})
        except Exception:  # noqa: BLE001 - an expression the planner / printer cannot take: host path
            return None
        _logger.info("post-step hook of %s runs on the device as `c <- %s`", solver.pde.__class__.__name__, expr)

        def post_step(arr: DeviceArray, t: float) -> DeviceArray:
            # IN PLACE: the pass is pointwise (no operators: checked above), every cell is read as the centre value only by the
            # thread that then writes it.  (Round 4 wrote into a recycled "spare" array and returned that: across stepper calls the
            # spare could be the caller's own `state_data`, i.e. the stepper's next output buffer - `cur is nxt`, an in-place
            # stencil sweep; ADVICE r4 high.  The hook now never hands out an array the stepper does not already hold as `cur`.)
            erhs.apply(arr, arr, "rate", 0.0, float(t))
            return arr

        post_step.on_device = True  # type: ignore[attr-defined]
        post_step.expression = expr  # type: ignore[attr-defined]
        return post_step

    def make_inner_stepper(self, solver, state):
        """Device-level stepper ``(state: DeviceArray, t_start, t_end) -> (DeviceArray, t_last)``.

        Fixed steps follow ``pde/backends/numba/_solvers.py:93-118``; the adaptive loop follows
        ``:240-281`` with ``_make_dt_adjuster`` (``pde/solvers/base.py:559-592``).
        """
        from .solvers import make_dt_adjuster

        post_step = self._make_host_post_step(solver, state)
        add_noise = self._make_noise_step(solver, state)
        if add_noise is not None:
            # Euler-Maruyama: deterministic Euler step, noise increment, then the hook (pde/solvers/euler.py:120-141)
            hook = post_step

            def post_step(arr, t, prev=None, _hook=hook):   # noqa: E306
                add_noise(arr, prev, t)      # (`prev`: the state before the step - a variance that depends on the field reads it)
                return arr if _hook is None else _hook(arr, t)

            post_step.wants_prev = True   # type: ignore[attr-defined]
        solver_name = solver.__class__.__name__
        if solver_name == "MilsteinSolver" and add_noise is None:
            solver_name = "EulerSolver"     # a deterministic equation: the Euler steps of its base class (pde/solvers/milstein.py:29)
        if solver_name not in {"EulerSolver", "RungeKuttaSolver", "ExplicitSolver", "AdamsBashforthSolver", "MilsteinSolver"}:
            msg = f"Backend `{self.name}` does not support solver {solver_name}"
            raise NotImplementedError(msg)
        if post_step is not None and solver_name == "AdamsBashforthSolver":
            msg = f"Backend `{self.name}` does not support post-step hooks with {solver_name}"
            raise NotImplementedError(msg)
        try:
            if np.dtype(state.dtype).kind == "c":
                # complex states: the equation as a real system of the parts through the run-time compiled passes (pde_hip/complex_expr.py)
                if add_noise is not None:
                    msg = f"Backend `{self.name}` does not support noise on complex fields"
                    raise RuntimeError(msg)
                msg = "complex state"
                raise NotImplementedError(msg)
            spec = self.make_rhs_spec(solver.pde, state)
        except NotImplementedError as err:
            if solver_name == "AdamsBashforthSolver":
                # expression PDEs (and complex states): the same two-step scheme around the run-time compiled right-hand side
                return self._make_adams_bashforth_expression_stepper(solver, self.make_expression_rhs(solver.pde, state))
            try:
                return self._make_expression_stepper(solver, state, post_step=post_step)   # generic expression PDE
            except NotImplementedError as err2:
                if any(c.__name__ in ("DiffusionPDE", "CahnHilliardPDE") for c in type(solver.pde).__mro__):
                    raise err from err2    # the reason the class right-hand side was refused is the informative one
                raise
        if post_step is not None:
            # the hook runs on the host between steps: the steps are driven from here, one sweep each
            return self._make_expression_stepper(solver, state, SpecRhs(self, spec), post_step=post_step)
        if spec.host_time_dependent:
            # faces given as Python functions: their coefficient arrays come from the host before every right-hand side, so the
            # steps are driven from here.  (Expression faces are refreshed on the device inside the C loops: spec.c.t below.)
            if solver_name == "AdamsBashforthSolver":
                msg = f"Backend `{self.name}` does not support time-dependent boundary conditions with {solver_name}"
                raise NotImplementedError(msg)
            return self._make_expression_stepper(solver, state, SpecRhs(self, spec))
        if solver_name == "AdamsBashforthSolver":
            return self._make_adams_bashforth_stepper(solver, spec)
        info, lib, stream = spec.info, self._lib, self.stream
        is_rk = solver_name == "RungeKuttaSolver"
        adaptive = bool(getattr(solver, "adaptive", False))
        work = [DeviceArray(info) for _ in range((7 if adaptive else 5) if is_rk else (3 if adaptive else 1))]   # adaptive Euler: rate, half step, scratch
        work_ptrs = ptr_array(work)
        if not adaptive:
            dt = float(solver.info["dt"])
            def fixed_stepper(state_data: DeviceArray, t_start: float, t_end: float):
                steps = max(1, round((t_end - t_start) / dt))
                spec.c.t = float(t_start)    # time of the first step: faces with explicit time dependence follow it inside the C loop
                if is_rk:
                    lib.rk4_run(info.ref, spec.ref, state_data.ptr, work_ptrs, dt, steps, stream)
                    result = state_data
                else:
                    res = C.c_void_p()
                    lib.euler_run(info.ref, spec.ref, state_data.ptr, work[0].ptr, dt, steps, C.byref(res), stream)
                    if res.value != state_data.ptr:
                        lib.memcpy_d2d(state_data.ptr, res.value, state_data.nbytes, stream)
                    result = state_data
                solver.info["steps"] += steps
                return result, t_start + (steps - 1) * dt + dt  # `t + dt` of the last iteration

            return fixed_stepper

        # adaptive stepping --------------------------------------------------------------------
        from .solvers import OnlineStatistics

        solver.info["dt_adaptive"] = True
        solver.info.setdefault("dt_statistics", OnlineStatistics())
        tolerance, dt_min = float(solver.tolerance), float(solver.dt_min)
        err_dev = DeviceScalar()
        ynew = DeviceArray(info)

        if os.environ.get("PDEHIP_ADAPTIVE_LOOP", "1") != "0":
            # The whole adaptive loop in ONE C call (the slab loop templates without a communicator and without neighbours = their
            # serial use): RKF45 attempts inside the generic loop of pde/backends/numba/_solvers.py:249-281 (`pdehip_slab_rkf45_run`),
            # or the reference's own adaptive Euler loop with the carried rate, :374-433 (`pdehip_slab_euler_adaptive_run`).  Stage
            # sequence, error norm, accept / reject, controller and step statistics run in C; the host reads 8 bytes per attempt.
            from .solvers import AdaptiveStatistics

            flags = C.c_int(0)
            lib.slab_flags_supported(info.ref, spec.ref, -1, -1, C.byref(flags))
            ctl = _abi.Adaptive()
            ctl.tolerance, ctl.dt_min, ctl.dt_max = tolerance, dt_min, float(solver.dt_max)
            solver.info["dt_statistics"] = AdaptiveStatistics(ctl)
            run = lib.slab_rkf45_run if is_rk else lib.slab_euler_adaptive_run

            def adaptive_loop(state_data: DeviceArray, t_start: float, t_end: float):
                ctl.t_start, ctl.t_end, ctl.dt = float(t_start), float(t_end), float(solver.info["dt"])
                before = int(ctl.steps)
                res = C.c_void_p()
                try:
                    run(None, info.ref, spec.ref, -1, -1, flags.value, state_data.ptr, ynew.ptr, work_ptrs, err_dev.ptr, C.byref(ctl), C.byref(res), stream)
                finally:
                    solver.info["steps"] += int(ctl.steps) - before
                    solver.info["attempts"] = int(ctl.attempts)      # accepted + rejected (not kept by the reference; bench.py prices an attempt)
                if res.value != state_data.ptr:
                    lib.memcpy_d2d(state_data.ptr, res.value, state_data.nbytes, stream)
                solver.info["dt"] = float(ctl.dt)
                return state_data, float(ctl.t_last)

            adaptive_loop.keepalive = (work, ynew, err_dev, spec)   # type: ignore[attr-defined]  (work_ptrs holds raw pointers only)
            return adaptive_loop

        # the same loops driven from Python (PDEHIP_ADAPTIVE_LOOP=0: a debugging aid): `_make_expression_stepper` holds them
        return self._make_expression_stepper(solver, state, SpecRhs(self, spec))

    def make_stepper(self, solver, state):
        """``stepper(state_field, t_start, t_end) -> t_last`` mutating ``state.data`` (base.py:728-755).

        The reference's device template moves the whole state over PCIe in both directions on EVERY call, i.e. at every
        tracker interrupt (``pde/backends/torch/backend.py:654-662``).  Here the state stays RESIDENT on the device between
        the calls of one stepper (config ``resident_state``, default on): the host copy of the field is refreshed only
        when somebody actually reads ``state.data`` (a tracker that stores or plots, the caller after the run), and the
        device copy is refreshed only after such an access (the view handed out is writable).  A run with ``tracker=None``
        or progress-only trackers uploads once and downloads once.  See :class:`ResidentState`.
        """
        inner = self.make_inner_stepper(solver, state)
        is_complex = np.dtype(state.dtype).kind == "c"     # complex states: planar (re, im) pairs of the real type on the device
        info = self.grid_info(state.grid, real_dtype_of(state.dtype))
        comp_shape = tuple(np.shape(state.data))[: np.ndim(state.data) - len(info.shape)] + ((2,) if is_complex else ())
        # a FieldCollection hands out its sub-fields as separate objects viewing the same memory: reads of `state[0].data`
        # cannot be intercepted, so collections take the plain upload / download per call
        resident = bool(_config_get(getattr(self, "config", None), "resident_state", True)) and state.__class__.__name__ != "FieldCollection"
        dev_state = DeviceArray(info, comp_shape, complex_pairs=is_complex)
        if not resident:

            def stepper(state_field, t_start: float, t_end: float) -> float:
                dev_state.set_valid(state_field.data, self.stream)
                result, t_last = inner(dev_state, t_start, t_end)
                result.get_valid(out=state_field.data, stream=self.stream)
                return t_last

            return stepper

        def resident_stepper(state_field, t_start: float, t_end: float) -> float:
            link = ResidentState.attach(state_field, dev_state, self)
            link.push()                                   # uploads only if the host copy may have changed
            try:
                result, t_last = inner(dev_state, t_start, t_end)
                if result is not dev_state:               # steppers hand back the array they were given; be safe
                    self._lib.memcpy_d2d(dev_state.ptr, result.ptr, dev_state.nbytes, self.stream)
            finally:
                link.device_advanced()                    # also when a post-step hook ends the run (StopIteration)
            return t_last

        resident_stepper.device_state = dev_state  # type: ignore[attr-defined]
        return resident_stepper
