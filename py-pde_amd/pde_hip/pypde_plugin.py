"""Registration of the hip backend with a real py-pde installation.

Usage (py-pde installed, libpdehip built)::

    import pde
    import pde_hip.pypde_plugin          # registers backend "hip" with pde.backends.backend_registry
    field.laplace(bc, backend="hip")
    eq.solve(state, t_range=10, dt=0.1, backend="hip", solver="euler")
    eq.solve(..., backend="hip:2")        # device 2

Mechanism (``pde/backends/registry.py:57-84``, ``:101-230``; pattern of
``pde/backends/torch/__init__.py:11-20``): ``backend_registry.register_package("hip",
"pde_hip.pypde_plugin", config=[...])`` declares the package lazily; importing this module defines
``HipBackend(HipBackendMixin, pde.backends.base.BackendBase)``, registers the Cartesian operators on
it and calls ``backend_registry.register_class("hip", HipBackend)``.  The mixin duck-types py-pde's
grids, ``BoundariesList`` and PDE / solver objects, so no py-pde class is re-implemented here.

Importing this module without py-pde raises ``ImportError``.  Constructing the backend never touches
the HIP runtime (``grid.operators`` instantiates every registered backend and only tolerates
``ImportError``, ``pde/backends/registry.py:241-245``): the first compute call selects the device and
raises ``RuntimeError`` when the library or a GPU is missing — there is no CPU fallback.
"""

from __future__ import annotations

try:
    import pde  # noqa: F401
    from pde.backends import backend_registry
    from pde.backends.base import BackendBase
    from pde.grids.cartesian import CartesianGrid
    from pde.solvers.base import AdaptiveSolverBase
    from pde.tools.config import Parameter
    from pde.trackers import trackers as _trackers
except ImportError as err:  # pragma: no cover - exercised only without py-pde
    msg = "pde_hip.pypde_plugin needs py-pde (`import pde` failed)"
    raise ImportError(msg) from err

import numpy as np

from . import operators as _operators
from .backend import HipBackendMixin
from .device import DeviceArray

DEFAULT_CONFIG = {
    "device": Parameter(value=-1, cls=int, description="Index of the HIP device (MI355X) the backend runs on; `hip:<n>` overrides it. "
                        "-1: the process default (LOCAL_RANK under a one-process-per-GPU launcher, else 0)."),
    "fastmath": Parameter(
        value=False,
        cls=bool,
        description="Compile-time arithmetic of the stencil kernels.  False (default): every rounding of the reference's expression order - "
        "results bit-identical to the numpy / torch-CPU backends.  True: the same kernels with FMA contraction, like the numba backend under "
        "its default `backend.numba.fastmath` - fewer fp64 operations per cell, results within 1e-10 (relative) of the exact build.",
    ),
    "resident_state": Parameter(
        value=True,
        cls=bool,
        description="Keep the state on the device between the calls of one stepper: tracker interrupts only download it, "
        "and it is uploaded again only when the host copy was modified in between.",
    ),
}


class HipBackend(HipBackendMixin, BackendBase):
    """py-pde backend running the finite-difference / explicit-stepper hot path on MI355X."""

    def __init__(self, config=None, *, name: str = "hip", device: int | None = None):
        BackendBase.__init__(self, config, name=name)
        if device is None and "device" in self.config and int(self.config["device"]) >= 0:
            device = int(self.config["device"])
        self._hip_init(device)

    @classmethod
    def from_args(cls, config, args: str = "", *, name: str | None = None):
        """``get_backend("hip:2")`` selects device 2 (pde/backends/registry.py:164-186)."""
        try:
            device = int(args) if args else None
        except ValueError:
            msg = f"hip backend: device index expected after the colon, got `{args}`"
            raise ValueError(msg) from None
        return cls(config, name=name or (f"hip:{args}" if args else "hip"), device=device)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(name={self.name!r}, device={self.device})"

    # py-pde hands native <-> numpy conversion only the array; the geometry is taken from the
    # operator/stepper closures, so plain arrays are accepted where the grid is implied
    def native_to_numpy(self, value):
        if isinstance(value, DeviceArray):
            return value.get_valid(stream=self.stream)
        return value

    def _apply_operator(self, func, *values: np.ndarray, out: np.ndarray, **kwargs) -> None:
        HipBackendMixin._apply_operator(self, func, *values, out=out, grid=getattr(func, "grid", None), **kwargs)


_operators.register_all(HipBackend, CartesianGrid)


def _install_evolution_rate_dispatch() -> None:
    """``pde.PDE.make_evolution_rate(state, backend)`` assembles the right-hand side from backend operators and
    ``backend.make_expression_function`` and wraps it per ``backend.implementation`` - with branches for "numpy" / "numba" / "jax" /
    "torch" only (pde/pdes/pde.py:469-494: any other implementation raises NotImplementedError).  This backend plans the WHOLE
    right-hand side at once (operators, user functions traced symbolically, conditions) in ``make_pde_rhs``; the three lines a maintainer
    would add there ("PDE.make_evolution_rate") are applied here at import time when the installed py-pde lacks them."""
    from pde.backends import get_backend
    from pde.pdes.pde import PDE

    if getattr(PDE.make_evolution_rate, "_hip_dispatch", False):
        return
    original = PDE.make_evolution_rate

    import functools
    import inspect

    signature = inspect.signature(original)

    @functools.wraps(original)
    def make_evolution_rate(self, *args, **kwargs):
        #  the call is passed on exactly as it came - whatever parameters, keywords and defaults the installed py-pde has;
        # only a backend argument that IS this backend (an object or the registered name "hip") is intercepted
        try:
            bound = signature.bind(self, *args, **kwargs)
            bound.apply_defaults()
            backend = bound.arguments.get("backend")
            state = bound.arguments.get("state")
        except TypeError:
            return original(self, *args, **kwargs)    # let py-pde raise its own error for a malformed call
        is_hip = getattr(backend, "implementation", None) == "hip" or backend == "hip"
        if not is_hip or state is None:
            return original(self, *args, **kwargs)
        return get_backend(backend).make_pde_rhs(self, state)

    make_evolution_rate._hip_dispatch = True  # type: ignore[attr-defined]
    PDE.make_evolution_rate = make_evolution_rate  # type: ignore[method-assign]


_install_evolution_rate_dispatch()


class _Statistics(dict):
    """Step-size statistics gathered by the C loop, with the one method py-pde's controller uses
    (``OnlineStatistics.to_dict``, pde/solvers/controller.py:285-287)."""

    def to_dict(self) -> dict:
        return dict(self)


class HipSlabSolver(AdaptiveSolverBase):
    """Explicit solver for runs with ONE PROCESS PER GPU (``python -m torch.distributed.run --nproc-per-node N script.py``):
    the counterpart of the reference's ``ExplicitMPISolver`` (``pde/solvers/explicit_mpi.py:24-226``, registered name
    ``"explicit_mpi"``) on the slab-parallel path of this backend (``pde_hip/distributed.py``).

    Every rank runs the same script with the same (replicated) initial state — there is no main / client split as with MPI:
    each stepper call cuts the state into axis-0 slabs, advances its own slab on its GPU with RCCL halo exchanges (one C call
    per call of the stepper), and all-gathers the result, so trackers on every rank see the full field.

        eq.solve(state, t_range=10, dt=0.1, solver="hip_slab", backend="hip")                       # Euler
        eq.solve(state, t_range=10, solver="hip_slab", scheme="runge-kutta", adaptive=True, backend="hip")
    """

    name = "hip_slab"

    def __init__(self, pde, scheme: str = "euler", *, backend="hip", adaptive: bool = False, tolerance: float = 1e-4, decomposition="slab",
                 gather: str = "all"):
        """``decomposition``: ``"slab"`` (axis-0 slabs: contiguous faces, two steps per sweep), ``"auto"`` (blocks by the reference's
        rule, pde/grids/_mesh.py:59-93: 2 x 2 x 2 for a cubic grid on 8 ranks) or a list of blocks per axis (``GridMesh.from_grid``).

        ``gather``: who receives the field when a stepper call ends (tracker interrupts, end of the run).  ``"all"`` (default): every rank -
        trackers run on every rank and see the same field, so they take the same decisions; raw buffers, one broadcast per part.  ``"root"``:
        rank 0 only, like the main node of the reference's ``ExplicitMPISolver`` (``GridMesh.combine_field_data_mpi``, pde/grids/_mesh.py:593-615)
        - every part crosses the control plane once (total traffic: one field per interrupt instead of N); the state of the other ranks carries
        their OWN part only, ``eq.solve`` returns that there, and trackers that look at the data (plots, storage, steady-state) belong on rank 0
        - trackers that can END a run must not be data-dependent on the other ranks."""
        super().__init__(pde, backend=backend, adaptive=adaptive, tolerance=tolerance)
        self.decomposition = decomposition
        if gather not in {"all", "root"}:
            msg = f"Unknown gather mode `{gather}` (all, root)"
            raise ValueError(msg)
        self.gather = gather
        if scheme in {"rk", "rk45", "runge-kutta"}:
            scheme = "runge-kutta"
        if scheme not in {"euler", "runge-kutta"}:
            msg = f"Unknown scheme `{scheme}` (euler, runge-kutta)"
            raise ValueError(msg)
        self.scheme = scheme

    def _collect(self, stepper, arr, state_field) -> None:
        """The result of a stepper call into ``state_field.data`` (see ``gather``): straight into its memory where the dtype and layout allow."""
        data = state_field.data
        root = 0 if self.gather == "root" else None
        direct = isinstance(data, np.ndarray) and data.flags.c_contiguous and data.flags.writeable and data.dtype == np.dtype(stepper.dtype)
        full = stepper.gather(arr, root=root, out=data if direct else None)
        if not direct:
            if full is not None:
                state_field.data[...] = full
            else:   # (a rank that receives nothing keeps its own part current)
                mesh = stepper.mesh
                lead = data.ndim - len(state_field.grid.shape)
                lo = list(mesh.lo) if isinstance(mesh.lo, (list, tuple)) else [mesh.lo]
                hi = list(mesh.hi) if isinstance(mesh.hi, (list, tuple)) else [mesh.hi]
                box = (slice(None),) * lead + tuple(slice(a, b) for a, b in zip(lo, hi))
                state_field.data[box] = arr.get_valid(stream=stepper.stream) if hasattr(arr, "get_valid") else stepper.gather_local(arr)

    def make_stepper(self, state, dt: float | None = None):
        from . import _abi
        from .distributed import SlabStepper

        if dt is None:
            dt = self.dt_default
        self.info.update(dt=float(dt), steps=0, dt_adaptive=bool(self.adaptive), stochastic=False, scheme=self.scheme, post_step_data=None)
        self._select_backend(state)
        # a post-step hook runs between the steps, on the host, on the box of each rank - the semantics of the reference's MPI solver
        # ("post_step_hook can only be used to do local modifications", pde/solvers/explicit_mpi.py:43-49): never inside the fused C loops
        try:
            self.pde.make_post_step_hook(state, backend="numpy")
            has_hook = True
        except NotImplementedError:
            has_hook = False                       # no hook defined: the normal case (pde/pdes/base.py:160-208)
        device = getattr(self.backend, "_device_request", None)
        # Diffusion / Cahn-Hilliard (classes or expressions of that form) on one scalar field: the fused loops (one C call per stepper
        # call); every other expression PDE - systems of scalar fields included -: its run-time compiled passes on the box of each
        # rank with a ghost exchange before every pass that applies operators (pde_hip.distributed.DecomposedExpressionStepper)
        from .distributed import SlabStepper

        try:
            if getattr(self.pde, "is_sde", False) or has_hook:
                msg = "the fused slab loops are deterministic and run without hooks"     # (both act between the steps: the Python-level stepper)
                raise NotImplementedError(msg)
            if state.__class__.__name__ != "ScalarField":
                msg = "the fused slab loops take one ScalarField"
                raise NotImplementedError(msg)
            SlabStepper._describe(self.pde, state.grid)
        except NotImplementedError:
            return self._make_expression_stepper(state, float(dt), device, has_hook)
        blocks = self.decomposition != "slab"
        if blocks:
            # a decomposition that only cuts axis 0 IS the slab decomposition (always the case for 1-D grids): the slab loops take it
            from .distributed import default_control
            from .mesh import block_decomposition

            from .distributed import resolve_decomposition

            size = default_control().size
            dims = block_decomposition(state.grid.shape, size) if self.decomposition == "auto" else resolve_decomposition(self.decomposition, size)
            blocks = any(d > 1 for d in dims[1:])
        if blocks:
            from .distributed import BlockStepper

            stepper = BlockStepper(self.pde, state.grid, state.dtype, dims=dims, device=device)
            self.info["decomposition"] = list(stepper.dims)
        else:
            stepper = SlabStepper(self.pde, state.grid, state.dtype, device=device)
            self.info["decomposition"] = [stepper.size] + [1] * (state.grid.num_axes - 1)
        self.info["world_size"] = stepper.size
        cur, nxt = stepper.buf("state_a"), stepper.buf("state_b")
        ctl = _abi.Adaptive()
        ctl.tolerance, ctl.dt_min, ctl.dt_max, ctl.dt = float(self.tolerance), float(self.dt_min), float(self.dt_max), float(dt)
        dt_fixed = float(dt)

        def slab_stepper(state_field, t_start: float, t_end: float) -> float:
            a, b = cur, nxt
            if blocks:
                a.set_valid(np.ascontiguousarray(stepper.mesh.extract(state_field.data), dtype=state_field.dtype), stepper.stream)
            else:
                stepper.set_local(a, stepper.mesh.extract(state_field.data))
            if self.adaptive:
                ctl.t_start, ctl.t_end = float(t_start), float(t_end)
                before = int(ctl.steps)
                res = stepper.rkf45_run(a, b, ctl) if self.scheme == "runge-kutta" else stepper.euler_adaptive_run(a, b, ctl)
                self.info["steps"] += int(ctl.steps) - before
                self.info["dt"] = float(ctl.dt)
                self.info["dt_statistics"] = _Statistics(_abi.adaptive_statistics(ctl))   # the controller calls .to_dict()
                t_last = float(ctl.t_last)
            else:
                steps = max(1, round((t_end - t_start) / dt_fixed))
                if self.scheme == "euler":
                    res = stepper.euler_steps(a, b, dt_fixed, steps, t_start)
                else:
                    stepper.rk4_steps(a, dt_fixed, steps, t_start)
                    res = a
                self.info["steps"] += steps
                t_last = t_start + (steps - 1) * dt_fixed + dt_fixed
            self._collect(stepper, res, state_field)
            return t_last

        slab_stepper.slab = stepper  # type: ignore[attr-defined]
        return slab_stepper


def _local_post_step(self, stepper, state):
    """The PDE's post-step hook on the BOX of this rank as ``post_step(array, t) -> array`` (host round trip of the box per step, like
    the single-device ``_make_host_post_step``).  The hook is made for the sub-field of the box (py-pde classes on a sub-grid with
    the box's bounds), its data stays local (``info["post_step_data"]``; all of them after every stepper call in
    ``info["post_step_data_list"]``, pde/solvers/explicit_mpi.py:121-131, :219-223).  ``StopIteration`` on ANY rank ends the run on all
    of them (agreed over the control plane after every step - a rank that stopped alone would leave the others in an exchange)."""
    grid, mesh = state.grid, stepper.mesh
    nd = grid.num_axes
    lo = list(mesh.lo) if stepper.blocks else [mesh.lo] + [0] * (nd - 1)
    hi = list(mesh.hi) if stepper.blocks else [mesh.hi] + [int(n) for n in grid.shape[1:]]
    dx = grid.discretization
    bounds = [(grid.axes_bounds[a][0] + lo[a] * dx[a], grid.axes_bounds[a][0] + hi[a] * dx[a]) for a in range(nd)]
    periodic = [bool(grid.periodic[a]) and stepper.dims[a] == 1 for a in range(nd)]
    subgrid = CartesianGrid(bounds, [h - l for l, h in zip(lo, hi)], periodic=periodic)
    if state.__class__.__name__ == "FieldCollection":
        sub_state = state.__class__([f.__class__(subgrid, data=np.ascontiguousarray(mesh.extract(f.data))) for f in state])
    else:
        sub_state = state.__class__(subgrid, data=np.ascontiguousarray(mesh.extract(state.data)))
    hook, data = self.pde.make_post_step_hook(sub_state, backend="numpy")
    self.info["post_step_data"] = data
    control = stepper.control

    def post_step(arr, t: float):
        host = arr.get_valid(stream=stepper.stream)
        stopped = False
        try:
            result = hook(host, t, self.info["post_step_data"])
            if result is not None:
                host, self.info["post_step_data"] = result
        except StopIteration:
            stopped = True        # (what the hook left in the array is the final state of this box)
        arr.set_valid(np.asarray(host, dtype=arr.dtype), stepper.stream)
        if any(control.allgather(stopped)):
            raise StopIteration
        return arr

    return post_step


def _decomposed_expression_stepper(self, state, dt: float, device, has_hook: bool = False):
    """``HipSlabSolver.make_stepper`` for PDEs without a fused decomposed loop."""
    from .distributed import DecomposedExpressionStepper

    kinds = [f.__class__.__name__ for f in (list(state) if state.__class__.__name__ == "FieldCollection" else [state])]
    if any(k not in ("ScalarField", "VectorField", "Tensor2Field") for k in kinds):
        msg = "slab-parallel stepping supports scalar, vector and rank-2 tensor fields (or a FieldCollection of them)"
        raise NotImplementedError(msg)
    dims = self.decomposition if isinstance(self.decomposition, str) else [int(d) for d in self.decomposition]    # (-1 entries: resolved there)
    stepper = DecomposedExpressionStepper(self.pde, state, dims=dims, device=device)
    self.info["decomposition"], self.info["world_size"] = list(stepper.dims), stepper.size
    post_step = _local_post_step(self, stepper, state) if has_hook else None
    step, sinfo = stepper.make_stepper(self.scheme, dt, adaptive=bool(self.adaptive), tolerance=float(self.tolerance), dt_min=float(self.dt_min),
                                       dt_max=float(self.dt_max), post_step=post_step)
    self.info["stochastic"] = bool(sinfo.get("stochastic", False))

    def expression_stepper(state_field, t_start: float, t_end: float) -> float:
        arr = stepper.scatter(state_field.data)
        before = int(sinfo["steps"])
        try:
            arr, t_last = step(arr, float(t_start), float(t_end))
        finally:
            # (also when a hook ended the run with StopIteration: every rank holds the whole field the hooks left behind)
            self.info["steps"] += int(sinfo["steps"]) - before
            self._collect(stepper, arr, state_field)
            if has_hook:
                self.info["post_step_data_list"] = stepper.control.allgather(self.info["post_step_data"])
        self.info["dt"], self.info["stochastic"] = float(sinfo["dt"]), bool(sinfo.get("stochastic", False))
        if "dt_statistics" in sinfo:
            self.info["dt_statistics"] = sinfo["dt_statistics"]
        return t_last

    expression_stepper.slab = stepper  # type: ignore[attr-defined]
    return expression_stepper


HipSlabSolver._make_expression_stepper = _decomposed_expression_stepper


class HipConsistencyTracker(_trackers.ConsistencyTracker):
    """``ConsistencyTracker`` (pde/trackers/trackers.py:974-1003) that does not pull the state to the host: while the state is
    resident on the device (between the stepper calls of a ``backend="hip"`` run) the finiteness check is a device reduction
    (``pdehip_count_nonfinite``).  Use ``tracker=["progress", "hip_consistency"]`` or an instance of this class."""

    name = "hip_consistency"

    def handle(self, field, t: float) -> None:
        link = getattr(field, "__dict__", {}).get("_hip_link")
        if link is None:
            return super().handle(field, t)
        if not link.backend.make_finite_check()(field):
            msg = "Field was not finite"
            raise StopIteration(msg)
        return None


def register() -> None:
    """(Re-)register the backend class with py-pde's registry; idempotent."""
    if "hip" not in backend_registry._packages:
        backend_registry.register_package("hip", __name__, config=DEFAULT_CONFIG)
    backend_registry.register_class("hip", HipBackend)


register()
