"""PDE definitions on the hot path: Diffusion, Cahn–Hilliard and the expression class ``PDE``.

Mirror of ``pde/pdes/base.py:355-565`` (``solve``, ``make_pde_rhs``), ``pde/pdes/diffusion.py:20-123``,
``pde/pdes/cahn_hilliard.py:20-124`` and the subset of ``pde/pdes/pde.py`` whose right-hand sides
map onto the fused device kernels.  ``evolution_rate`` is evaluated on the device.
"""

from __future__ import annotations

from typing import Any

from .fields import ScalarField


class PDEBase:
    explicit_time_dependence = False
    complex_valued = False
    is_sde = False

    def __init__(self):
        self.diagnostics: dict[str, Any] = {}

    def make_pde_rhs(self, state, backend="hip"):
        """``rhs(state_native, t)`` from the backend (pdes/base.py:402-427)."""
        from .backend import get_backend

        return get_backend(backend).make_pde_rhs(self, state)

    def evolution_rate(self, state: ScalarField, t: float = 0, *, backend="hip") -> ScalarField:
        """Right-hand side for ``state`` as a new field (evaluated by the device kernels)."""
        from .backend import get_backend

        b = get_backend(backend)
        rhs = b.make_pde_rhs(self, state)
        native = b.numpy_to_native(state.data, grid=state.grid)
        return ScalarField(state.grid, b.native_to_numpy(rhs(native, t)), label="evolution rate")

    def solve(self, state, t_range, dt: float | None = None, tracker=None, *, solver="euler", ret_info: bool = False,
              backend="hip", interval: float | None = None, **kwargs):
        """Solve the PDE (pdes/base.py:451-565).  ``adaptive`` defaults to ``dt is None``."""
        from .solvers import Controller, SolverBase

        if isinstance(solver, str):
            if solver in {"euler", "explicit", "runge-kutta"}:   # pdes/base.py:532-535
                kwargs.setdefault("adaptive", dt is None)
            solver_obj = SolverBase.from_name(solver, pde=self, backend=backend, **kwargs)
        else:
            solver_obj = solver
        controller = Controller(solver_obj, t_range=t_range, tracker=tracker, interval=interval)
        final = controller.run(state, dt)
        self.diagnostics = controller.diagnostics
        if ret_info:
            return final, self.diagnostics
        return final


class DiffusionPDE(PDEBase):
    r""":math:`\partial_t c = D \nabla^2 c` (pdes/diffusion.py)."""

    default_bc = "auto_periodic_neumann"

    use_noise_variance = True
    use_noise_realization = False

    def __init__(self, diffusivity: float = 1, *, bc=None, noise: float = 0, rng=None):
        super().__init__()
        self.diffusivity = diffusivity
        self.noise = float(noise)          # variance of additive Gaussian white noise (pde/pdes/base.py:583-616)
        self.rng = rng
        self.bc = self.default_bc if bc is None else bc

    @property
    def is_sde(self) -> bool:
        return self.noise != 0

    @property
    def expression(self) -> str:
        return f"{self.diffusivity} * laplace(c)"


class CahnHilliardPDE(PDEBase):
    r""":math:`\partial_t c = \nabla^2(c^3 - c - \gamma \nabla^2 c)` (pdes/cahn_hilliard.py)."""

    default_bc_c = "auto_periodic_neumann"
    default_bc_mu = "auto_periodic_neumann"

    def __init__(self, interface_width: float = 1, *, bc_c=None, bc_mu=None):
        super().__init__()
        self.interface_width = interface_width
        self.bc_c = self.default_bc_c if bc_c is None else bc_c
        self.bc_mu = self.default_bc_mu if bc_mu is None else bc_mu

    @property
    def expression(self) -> str:
        return f"laplace(c**3 - c - {self.interface_width} * laplace(c))"


class _ScalarClassPDE(PDEBase):
    """Parameter holder of the reference's other scalar PDE classes; the backend builds their right-hand sides from the
    attributes (``backend.class_expressions``), so these mirrors carry nothing else."""

    default_bc = "auto_periodic_neumann"

    def _conditions(self, bc, bc_lap=None) -> None:
        self.bc = self.default_bc if bc is None else bc
        self.bc_lap = self.default_bc if bc_lap is None else bc_lap


class AllenCahnPDE(_ScalarClassPDE):
    r""":math:`\partial_t c = M (\gamma \nabla^2 c - c^3 + c)` (pdes/allen_cahn.py:20-128)."""

    def __init__(self, interface_width: float = 1, mobility: float = 1, *, bc=None):
        super().__init__()
        self.interface_width, self.mobility = interface_width, mobility
        self._conditions(bc)


class KPZInterfacePDE(_ScalarClassPDE):
    r""":math:`\partial_t h = \nu \nabla^2 h + \lambda |\nabla h|^2` (pdes/kpz_interface.py:21-137)."""

    def __init__(self, nu: float = 0.5, lmbda: float = 1, *, bc=None):
        super().__init__()
        self.nu, self.lmbda = nu, lmbda
        self._conditions(bc)


class KuramotoSivashinskyPDE(_ScalarClassPDE):
    r""":math:`\partial_t u = -\nu \nabla^4 u - \nabla^2 u - |\nabla u|^2 / 2` (pdes/kuramoto_sivashinsky.py:21-146)."""

    def __init__(self, nu: float = 1, *, bc=None, bc_lap=None):
        super().__init__()
        self.nu = nu
        self._conditions(bc, bc_lap)


class SwiftHohenbergPDE(_ScalarClassPDE):
    r""":math:`\partial_t c = [\epsilon - (k_c^2 + \nabla^2)^2] c + \delta c^2 - c^3` (pdes/swift_hohenberg.py:21-153)."""

    def __init__(self, rate: float = 0.1, kc2: float = 1.0, delta: float = 1.0, *, bc=None, bc_lap=None):
        super().__init__()
        self.rate, self.kc2, self.delta = rate, kc2, delta
        self._conditions(bc, bc_lap)


class WavePDE(_ScalarClassPDE):
    r""":math:`\partial_t u = v,\ \partial_t v = c^2 \nabla^2 u` on a collection (u, v) (pdes/wave.py:21-137)."""

    def __init__(self, speed: float = 1, *, bc=None):
        super().__init__()
        self.speed = speed
        self._conditions(bc)

    def get_initial_condition(self, u: ScalarField, v: ScalarField | None = None):
        from .fields import FieldCollection

        if v is None:
            v = ScalarField(u.grid, 0.0 * u.data, dtype=u.dtype)
        return FieldCollection([u, v])


class KleinGordonPDE(WavePDE):
    r""":math:`\partial_t u = v,\ \partial_t v = c^2 \nabla^2 u - m^2 u` (pdes/klein_gordon.py:21-160)."""

    def __init__(self, speed: float = 1, mass: float = 1, *, bc=None):
        super().__init__(speed, bc=bc)
        self.mass = mass


class PDE(PDEBase):
    """PDE given by an expression string per variable (pdes/pde.py).

    The hip backend recognises ``D*laplace(c)`` and ``laplace(c**3 - c - g*laplace(c))`` and runs
    them on the fused kernels; any other expression raises ``NotImplementedError`` (a run-time
    expression compiler is the next component).
    """

    use_noise_variance = True
    use_noise_realization = False

    def __init__(self, rhs: dict[str, str], *, bc="auto_periodic_neumann", bc_ops=None, consts=None, noise=0, rng=None, user_funcs=None,
                 post_step_hook=None):
        super().__init__()
        self.post_step_hook = post_step_hook          # `f(state_data, t) -> state_data` after every step (pde/pdes/pde.py:99-118, :671-706)
        self.user_funcs = dict(user_funcs or {})      # Python functions the expressions may call (traced symbolically by the backend)
        self.rhs = {k: str(v) for k, v in rhs.items()}
        self.variables = tuple(self.rhs)
        self.consts = dict(consts or {})
        # variance of additive Gaussian white noise: one number or one per field (pde/pdes/pde.py:266-281)
        import numpy as np

        self.noise = np.broadcast_to(np.asarray(noise, dtype=float), (len(self.variables),)).copy()
        self.rng = rng
        # boundary conditions per "variable:operator", wildcards allowed, default last (pde/pdes/pde.py:232-264)
        if bc_ops is not None and not isinstance(bc_ops, dict):
            msg = f"`bc_ops` must be a dictionary, but got {type(bc_ops)}"
            raise TypeError(msg)
        self.bc, self.bc_ops = bc, dict(bc_ops or {})   # kept for convenience; the backend reads `bcs` like on pde.PDE
        table = dict(bc_ops or {})
        table["*:*"] = bc
        self.bcs: dict[str, Any] = {}
        for key_str, value in table.items():
            parts = key_str.replace(".", ":").split(":")
            if len(parts) == 1:
                key = f"{self.variables[0]}:{key_str}"
            elif len(parts) == 2:
                key = ":".join(parts)
            else:
                msg = f'Cannot parse boundary condition "{key_str}"'
                raise ValueError(msg)
            self.bcs[key] = value

    @property
    def is_sde(self) -> bool:
        return bool((self.noise != 0).any())

    def make_post_step_hook(self, state, backend="numpy"):
        """``(hook(state_data, t, data) -> (state_data, data), initial data)`` around ``post_step_hook`` (pde/pdes/pde.py:671-706)."""
        if self.post_step_hook is None:
            msg = "`post_step_hook` not set"
            raise NotImplementedError(msg)
        user = self.post_step_hook

        def post_step_hook_impl(state_data, t, post_step_data):
            return user(state_data, t), None

        return post_step_hook_impl, None

    @property
    def complex_valued(self) -> bool:
        """The right-hand side contains the imaginary unit or a complex constant (pde/pdes/pde.py:206-216): the controller then
        evolves a complex state (pde/solvers/controller.py:430-432)."""
        import re

        import numpy as np

        if any(np.iscomplexobj(v) for v in self.consts.values()):
            return True
        return any(re.search(r"(?<![A-Za-z0-9_])I(?![A-Za-z0-9_])", str(e)) for e in self.rhs.values())

    @property
    def expressions(self) -> dict[str, str]:
        """Expressions after the shorthand replacement of the reference (pde/pdes/pde.py:47-53)."""
        return {var: str(e).replace("∇²", "laplace") for var, e in self.rhs.items()}

    @property
    def expression(self) -> str:
        return next(iter(self.rhs.values()))
