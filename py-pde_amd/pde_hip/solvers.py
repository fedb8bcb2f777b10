"""Explicit solvers and the controller loop, host side.

Mirror of the pieces of ``pde.solvers`` that sit directly above the backend boundary:
``SolverBase.make_stepper`` (``pde/solvers/base.py:298-332``), ``EulerSolver`` /
``RungeKuttaSolver`` (``pde/solvers/euler.py:40-289``, ``runge_kutta.py:24-156``), the time-step
controller ``_make_dt_adjuster`` (``base.py:533-594``), ``OnlineStatistics``
(``pde/tools/math.py:125-174``) and the interrupt loop of ``Controller._run_main_process``
(``pde/solvers/controller.py:146-298``).  The update rules themselves run on the device
(``HipBackendMixin.make_inner_stepper``); nothing here touches field data.
"""

from __future__ import annotations

import math
import time
from typing import Any, Callable

import numpy as np


class OnlineStatistics:
    """Running mean / variance (Welford), same interface as ``pde.tools.math.OnlineStatistics``."""

    def __init__(self):
        self.min, self.max = math.inf, -math.inf
        self.mean, self._mean2, self.count = 0.0, 0.0, 0

    def add(self, value: float) -> None:
        self.min, self.max = min(self.min, value), max(self.max, value)
        delta = value - self.mean
        self.count += 1
        self.mean += delta / self.count
        self._mean2 += delta * (value - self.mean)

    @property
    def var(self) -> float:
        return self._mean2 / (self.count - 1) if self.count >= 2 else math.nan

    @property
    def std(self) -> float:
        return math.sqrt(self.var)

    def to_dict(self) -> dict[str, Any]:
        return {"min": self.min, "max": self.max, "mean": self.mean, "std": self.std, "count": self.count}


class AdaptiveStatistics:
    """Statistics of the accepted step sizes kept by the C loops (``pdehip_adaptive_t``: Welford sums like
    ``pde.tools.math.OnlineStatistics``, pde/tools/math.py:125-174), read through the interface the controller and users
    see: ``count`` / ``min`` / ``max`` / ``mean`` / ``std`` / ``to_dict()`` (pde/solvers/controller.py:285-287)."""

    def __init__(self, ctl):
        self._ctl = ctl

    def to_dict(self) -> dict[str, Any]:
        from ._abi import adaptive_statistics

        return adaptive_statistics(self._ctl)

    def __getattr__(self, name):
        if name in {"count", "min", "max", "mean", "std"}:
            return self.to_dict()[name]
        raise AttributeError(name)

    @property
    def var(self) -> float:
        std = self.to_dict()["std"]
        return std * std


def make_dt_adjuster(dt_min: float, dt_max: float) -> Callable[[float, float], float]:
    """``adjust_dt(dt, error_rel)`` keeping ``error_rel`` near 1 (solvers/base.py:559-592)."""

    def adjust_dt(dt: float, error_rel: float) -> float:
        if error_rel < 0.00057665:
            dt *= 4.0
        elif math.isnan(error_rel):
            dt *= 0.25
        else:
            dt *= max(0.9 * error_rel**-0.2, 0.1)
        if dt > dt_max:
            dt = dt_max
        elif dt < dt_min:
            if math.isnan(error_rel):
                msg = f"Encountered NaN even though dt < {dt_min}"
                raise RuntimeError(msg)
            msg = f"Time step below {dt_min}"
            raise RuntimeError(msg)
        return dt

    return adjust_dt


class SolverBase:
    """Base class of the explicit solvers (solvers/base.py:50-332)."""

    name = "base"
    dt_default = 1e-3
    _subclasses: dict[str, type["SolverBase"]] = {}

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        SolverBase._subclasses[cls.name] = cls

    def __init__(self, pde, *, backend="hip"):
        from .backend import get_backend

        self.pde = pde
        self.backend = get_backend(backend)
        self.mpi_run = False
        self.info: dict[str, Any] = {"class": self.__class__.__name__, "pde_class": pde.__class__.__name__}

    adaptive = False

    @classmethod
    def from_name(cls, name: str, pde, **kwargs) -> "SolverBase":
        try:
            solver_cls = cls._subclasses[name]
        except KeyError:
            msg = f"Unknown solver method '{name}'. Registered solvers are {sorted(cls._subclasses)}"
            raise ValueError(msg) from None
        return solver_cls(pde, **kwargs)

    def make_stepper(self, state, dt: float | None = None):
        """``stepper(state, t_start, t_end) -> t_last`` (solvers/base.py:298-332)."""
        if dt is None:
            dt = self.dt_default
        self.info["dt"] = float(dt)
        self.info["dt_adaptive"] = bool(self.adaptive)
        self.info["steps"] = 0
        self.info["stochastic"] = False
        self.info["backend"] = self.backend.info
        self.info["post_step_data"] = None
        return self.backend.make_stepper(self, state)


class AdaptiveSolverBase(SolverBase):
    dt_min = 1e-10
    dt_max = 1e10

    def __init__(self, pde, *, backend="hip", adaptive: bool = False, tolerance: float = 1e-4):
        super().__init__(pde, backend=backend)
        self.adaptive = bool(adaptive)
        self.tolerance = float(tolerance)


class EulerSolver(AdaptiveSolverBase):
    """Explicit Euler: ``y += dt * f(y, t)`` (solvers/euler.py:172-175)."""

    name = "euler"


class ExplicitSolver(EulerSolver):
    """Deprecated alias kept by the reference (solvers/euler.py:292-326)."""

    name = "explicit"


class RungeKuttaSolver(AdaptiveSolverBase):
    """RK4 with fixed steps, RKF45 when adaptive (solvers/runge_kutta.py:24-156)."""

    name = "runge-kutta"


class AdamsBashforthSolver(SolverBase):
    """Explicit two-step Adams-Bashforth: ``y += dt * (1.5 f(y_n) - 0.5 f(y_{n-1}))`` (solvers/adams_bashforth.py:17-75)."""

    name = "adams-bashforth"


class Controller:
    """Advance a state over ``t_range``, interrupting for trackers (solvers/controller.py:146-298).

    ``tracker`` is ``None`` or a callable ``tracker(state, t)``; ``interval`` is the time between
    two tracker calls.  With ``tracker=None`` the stepper is called once for the whole range.
    """

    def __init__(self, solver: SolverBase, t_range, tracker=None, interval: float | None = None):
        self.solver = solver
        self.t_range = (0.0, float(t_range)) if np.isscalar(t_range) else (float(t_range[0]), float(t_range[1]))
        self.tracker = tracker
        self.interval = interval
        self.info: dict[str, Any] = {}
        self.diagnostics: dict[str, Any] = {}

    def run(self, initial_state, dt: float | None = None):
        state = initial_state.copy()
        if bool(getattr(self.solver.pde, "complex_valued", False)) and state.dtype.kind != "c":
            state = state.copy(dtype=complex)        # pde/solvers/controller.py:430-432
        t_start, t_end = self.t_range
        t0 = time.perf_counter()
        stepper = self.solver.make_stepper(state, dt)
        t_compile = time.perf_counter() - t0
        dt_used = self.solver.info["dt"]
        atol = 1e-6 * dt_used  # controller.py:199-206
        t = t_start
        t_solver = 0.0
        while t < t_end - atol:
            t_next = t_end
            if self.tracker is not None:
                self.tracker(state, t)
                if self.interval:
                    t_next = min(t_end, t + self.interval)
            t1 = time.perf_counter()
            t = stepper(state, t, t_next)
            t_solver += time.perf_counter() - t1
        if self.tracker is not None:
            self.tracker(state, t)
        info = dict(self.solver.info)
        if "dt_statistics" in info and hasattr(info["dt_statistics"], "to_dict"):
            info["dt_statistics"] = info["dt_statistics"].to_dict()
        self.info = {"t_final": t, "profiler": {"solver": t_solver, "compilation": t_compile}}
        self.diagnostics = {"controller": self.info, "solver": info}
        return state
