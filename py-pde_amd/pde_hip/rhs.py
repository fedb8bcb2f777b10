"""Right-hand sides: which PDE classes map onto the fused kernels (``RhsSpec``), their expressions and boundary-condition tables, and the
planning methods of the backend (``make_rhs_spec`` / ``make_pde_rhs`` / ``make_expression_rhs``) as :class:`RhsPlanningMixin`.  Split from
``backend.py`` in round 6 (no behaviour change).

Reference: ``pde/pdes/diffusion.py:99-123``, ``pde/pdes/cahn_hilliard.py:95-124``, ``pde/pdes/pde.py:299-499``, ``pde/backends/base.py:634-651``.
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
from .faces import HostSetterTable, convert_bcs, real_dtype_of

_logger = logging.getLogger("pde_hip.backend")


# ---------------------------------------------------------------------------------------------
# right hand sides the fused steppers know
# ---------------------------------------------------------------------------------------------
# methods that define the right-hand side of a PDE class (pde/pdes/base.py:211-449, pde/pdes/pde.py:636-900): a user subclass that
# overrides ANY of them is a different equation - the reference honours the override through `eq.make_evolution_rate`, this backend
# maps the class onto a built-in kernel and must therefore refuse it
_RHS_METHODS = ("evolution_rate", "make_evolution_rate", "make_pde_rhs", "_make_pde_rhs_numba", "_make_pde_rhs_numba_cached",
                "_make_pde_rhs_collection_numba", "_make_pde_rhs_collection_torch", "_make_pde_rhs_collection_jax",
                "_compile_rhs_single", "_add_operators_to_expr", "_prepare_cache", "expression", "expressions")


def known_pde_class(eq, names) -> type | None:
    """The class of ``eq`` or the nearest base class whose NAME is in ``names`` (the reference's and the mirror's classes both
    match) — provided no class between ``type(eq)`` and it redefines a method of the right-hand side.  A subclass that only
    adds e.g. a post-step hook is accepted; one that changes the equation raises ``NotImplementedError`` (``backend="auto"``
    treats that as "try the next backend", pde/pdes/base.py:383-400)."""
    mro = type(eq).__mro__
    for i, cls in enumerate(mro):
        if cls.__name__ in names:
            for sub in mro[:i]:
                changed = [m for m in _RHS_METHODS if m in vars(sub)]
                if changed:
                    msg = (f"hip backend: {sub.__name__} overrides {', '.join(changed)} of {cls.__name__}; user-defined right-hand sides "
                           "in Python cannot run on the device (the built-in kernel of the base class would silently ignore the override)")
                    raise NotImplementedError(msg)
            return cls
    return None


def pde_kind(eq) -> str:
    """``"DiffusionPDE"`` / ``"CahnHilliardPDE"`` / ``"PDE"`` for objects of these classes OR subclasses that leave the
    right-hand side alone (see :func:`known_pde_class`); otherwise the object's own class name."""
    cls = known_pde_class(eq, {"DiffusionPDE", "CahnHilliardPDE", "PDE"})
    return cls.__name__ if cls is not None else eq.__class__.__name__


def class_expressions(eq):
    """The built-in PDE classes of the reference beyond Diffusion / Cahn-Hilliard as expression systems for the run-time
    specialised kernels: ``(rhs: {variable: expression}, consts, bcs: {(variable, operator name): condition}, aliases)`` or None.
    Formulas and the condition each (nested) operator takes are those of the classes' ``evolution_rate``:
    AllenCahnPDE pde/pdes/allen_cahn.py:98-100, KPZInterfacePDE kpz_interface.py:104-107, KuramotoSivashinskyPDE
    kuramoto_sivashinsky.py:106-111, SwiftHohenbergPDE swift_hohenberg.py:104-113, WavePDE wave.py:106-109, KleinGordonPDE
    klein_gordon.py:124-127.  Matched by class name along the MRO like :func:`pde_kind` (subclasses that redefine the
    right-hand side are refused, :func:`known_pde_class`)."""
    base = known_pde_class(eq, {"AllenCahnPDE", "KPZInterfacePDE", "KuramotoSivashinskyPDE", "SwiftHohenbergPDE", "KleinGordonPDE", "WavePDE",
                                "CahnHilliardPDE", "DiffusionPDE"})
    if base is None:
        return None
    names = [cls.__name__ for cls in base.__mro__]
    outer = {"laplace_outer": "laplace"}
    if "CahnHilliardPDE" in names:
        # (pde/pdes/cahn_hilliard.py:115-122; the fused class right-hand side - RhsSpec - comes first: this form serves what it
        # refuses, e.g. conditions of mu that depend non-linearly on mu)
        return ({"c": "laplace_outer(c**3 - c - interface_width * laplace(c))"}, {"interface_width": float(eq.interface_width)},
                {("c", "laplace"): eq.bc_c, ("c", "laplace_outer"): eq.bc_mu}, outer)
    if "DiffusionPDE" in names:
        # (pde/pdes/diffusion.py:119-121; the fused class right-hand side comes first: this form serves the decomposed steppers
        # for schemes without a fused loop, e.g. adaptive Euler)
        return ({"c": "diffusivity * laplace(c)"}, {"diffusivity": float(eq.diffusivity)}, {("c", "laplace"): eq.bc}, {})
    if "AllenCahnPDE" in names:
        return ({"c": "mobility * (interface_width * laplace(c) - c**3 + c)"},
                {"mobility": float(eq.mobility), "interface_width": float(eq.interface_width)}, {("c", "laplace"): eq.bc}, {})
    if "KPZInterfacePDE" in names:
        return ({"c": "nu * laplace(c) + lmbda * gradient_squared(c)"}, {"nu": float(eq.nu), "lmbda": float(eq.lmbda)},
                {("c", "laplace"): eq.bc, ("c", "gradient_squared"): eq.bc}, {})
    if "KuramotoSivashinskyPDE" in names:
        # the form the reference's solvers use (make_evolution_rate, kuramoto_sivashinsky.py:139-144): the outer operator - and
        # with it the conditions `bc_lap` - is applied to MINUS the inner Laplacian; `evolution_rate` (:106-111) applies it to
        # the Laplacian itself, which differs for inhomogeneous `bc_lap`
        return ({"c": "-laplace(c) + nu * laplace_outer(-laplace(c)) - 0.5 * gradient_squared(c)"}, {"nu": float(eq.nu)},
                {("c", "laplace"): eq.bc, ("c", "gradient_squared"): eq.bc, ("c", "laplace_outer"): eq.bc_lap}, outer)
    if "SwiftHohenbergPDE" in names:
        return ({"c": "(rate - kc2**2) * c - 2 * kc2 * laplace(c) - laplace_outer(laplace(c)) + delta * c**2 - c**3"},
                {"rate": float(eq.rate), "kc2": float(eq.kc2), "delta": float(eq.delta)},
                {("c", "laplace"): eq.bc, ("c", "laplace_outer"): eq.bc_lap}, outer)
    if "KleinGordonPDE" in names:
        return ({"u": "v", "v": "speed**2 * laplace(u) - mass**2 * u"}, {"speed": float(eq.speed), "mass": float(eq.mass)},
                {("v", "laplace"): eq.bc}, {})
    if "WavePDE" in names:
        return ({"u": "v", "v": "speed**2 * laplace(u)"}, {"speed": float(eq.speed)}, {("v", "laplace"): eq.bc}, {})
    return None


class RhsSpec:
    """``pdehip_rhs_t`` + everything that must stay alive with it."""

    def __init__(self, kind: int, param: float, info: GridInfo, bc_c: FaceTable, bc_mu: FaceTable | None = None):
        self.kind, self.param, self.info = kind, float(param), info
        self.bc_c, self.bc_mu = bc_c, bc_mu
        self.c = _abi.RHS()
        self.c.kind = kind
        self.c.param = float(param)
        bc_c.copy_into(self.c.bc_c)
        self.mu = None
        if kind == _abi.RHS_CAHN_HILLIARD:
            assert bc_mu is not None
            bc_mu.copy_into(self.c.bc_mu)
            self.mu = DeviceArray(info)
            self.c.scratch_mu = self.mu.ptr
        # faces with explicit time dependence: ONE device program for both tables, run by every C entry point for the time of its
        # evaluation (`pdehip_rhs_t::bc_program`, `t`); faces given as Python functions stay on the host (`host_time_dependent`)
        # (... and faces that are not affine in the adjacent value: the program reads the input field of every evaluation)
        self.program = None
        if bc_mu is not None and getattr(bc_mu, "reads_value", False):
            msg = "hip backend: conditions of the chemical potential that depend non-linearly on it (mu is never stored between the two operators)"
            raise NotImplementedError(msg)
        if self.host_time_dependent and any(getattr(tb, "reads_value", False) for tb in (bc_c, bc_mu) if tb is not None):
            msg = "hip backend: conditions given as Python functions together with conditions that depend non-linearly on the field"
            raise NotImplementedError(msg)
        if self.time_dependent and not self.host_time_dependent:
            from .bc_expr import program_for

            self.program = program_for(require_device(), [self.bc_c, self.bc_mu], info)
            if self.program is not None:
                self.c.bc_program = self.program.ptr

    @property
    def time_dependent(self) -> bool:
        """Faces whose coefficient arrays must be refreshed when the time changes (expression BCs with `t`)."""
        return any(getattr(tb, "time_dependent", False) for tb in (self.bc_c, self.bc_mu) if tb is not None)

    @property
    def host_time_dependent(self) -> bool:
        """... and some of them are Python functions: refreshed from the host, which keeps the steps out of the C loops."""
        return any(getattr(tb, "host_only", False) for tb in (self.bc_c, self.bc_mu) if tb is not None)

    def update(self, t: float, stream=None) -> None:
        """Time of the next evaluation: the C entry points refresh the device-evaluated faces themselves (``self.c.t``); faces
        given as Python functions get their coefficient arrays from the host here (copied on ``stream``, the consumers' stream)."""
        self.c.t = float(t)
        if self.program is None:
            for tb in (self.bc_c, self.bc_mu):
                if tb is not None and getattr(tb, "time_dependent", False):
                    tb.update({"t": t}, stream=stream)

    @property
    def ref(self):
        return C.byref(self.c)


def pde_bcs_table(eq) -> dict[str, Any]:
    """``{"var:operator": bc}`` of an expression PDE in lookup order.

    The reference's ``pde.PDE`` stores exactly this as ``eq.bcs`` (``pde/pdes/pde.py:232-264``: the entries of
    ``bc_ops`` in insertion order, then ``"*:*"`` = ``bc``; keys without a variable get the first one).  Objects that
    only carry ``bc`` / ``bc_ops`` (older mirror instances) are normalised the same way.  Anything else is refused:
    silently falling back to default conditions would give wrong results.
    """
    bcs = getattr(eq, "bcs", None)
    if isinstance(bcs, dict):
        return bcs
    if not hasattr(eq, "bc"):
        msg = f"hip backend: cannot determine the boundary conditions of {eq.__class__.__name__} (no `bcs` / `bc` attribute)"
        raise NotImplementedError(msg)
    variables = list(getattr(eq, "rhs", {}))
    table: dict[str, Any] = {}
    for key, value in dict(getattr(eq, "bc_ops", None) or {}).items():
        parts = key.replace(".", ":").split(":")
        if len(parts) == 1:
            key = f"{variables[0]}:{key}"
        elif len(parts) != 2:
            msg = f'Cannot parse boundary condition "{key}"'
            raise ValueError(msg)
        else:
            key = ":".join(parts)
        table[key] = value
    table["*:*"] = eq.bc
    return table


def pde_bc_for(eq, var: str, operator: str):
    """Boundary condition the reference applies to ``operator`` in the equation of ``var``: the FIRST entry of
    ``eq.bcs`` whose variable and operator match, ``*`` being a wildcard (``pde/pdes/pde.py:329-343``); one condition
    per operator NAME, used for every (also nested) application of it."""
    for key, bc in pde_bcs_table(eq).items():
        bc_var, bc_func = key.split(":")
        if bc_var in (var, "*") and bc_func in (operator, "*"):
            return bc
    msg = f"Could not find suitable boundary condition for function `{operator}` applied in equation for `{var}`"
    raise RuntimeError(msg)


def pde_expression(eq, var: str) -> str:
    """Expression string of ``var`` after the reference's shorthand replacement (``pde/pdes/pde.py:47-53``, ``:195-201``)."""
    exprs = getattr(eq, "expressions", None)
    if isinstance(exprs, dict) and var in exprs:
        return str(exprs[var])
    return str(dict(eq.rhs)[var])


def _match_expression_rhs(expr_str: str, var: str, consts: dict[str, Any]) -> tuple[int, float] | None:
    """Recognise ``D*laplace(c)`` and ``laplace(c**3 - c - g*laplace(c))`` (SURVEY.md cfg 5)."""
    import sympy

    expr_str = expr_str.replace("∇²", "laplace").replace("^", "**")
    lap = sympy.Function("laplace")
    c = sympy.Symbol(var)
    local = {"laplace": lap, var: c}
    for k, v in consts.items():
        if np.isscalar(v):
            local[k] = sympy.Float(float(v))
    try:
        expr = sympy.sympify(expr_str, locals=local)
    except (sympy.SympifyError, SyntaxError, TypeError):
        return None
    D = sympy.Wild("D", exclude=[c, lap])
    m = expr.match(D * lap(c))
    if m is not None and m[D].is_number:
        return _abi.RHS_DIFFUSION, float(m[D])
    if isinstance(expr, lap) and len(expr.args) == 1:
        inner = sympy.expand(expr.args[0])
        g = sympy.Wild("g", exclude=[c, lap])
        m = inner.match(c**3 - c - g * lap(c))
        if m is not None and m[g].is_number:
            return _abi.RHS_CAHN_HILLIARD, float(m[g])
    return None


class SpecRhs:
    """A fused class right-hand side (:class:`RhsSpec`) behind the evaluator interface of
    :class:`~pde_hip.expr.ExpressionRhs`, for steppers driven from Python: every evaluation first refreshes the
    coefficient arrays of time-dependent faces (``args={"t": t}`` of the reference, ``pde/pdes/diffusion.py:119-121``)."""

    def __init__(self, backend, spec: RhsSpec):
        self.backend, self.spec, self.info, self.lib = backend, spec, spec.info, backend._lib

    def apply(self, state, out, wrap: str = "rate", dt: float = 0.0, t: float = 0.0) -> None:
        spec, st = self.spec, self.backend.stream
        spec.update(t, st)
        if wrap == "euler":
            res = C.c_void_p()
            self.lib.euler_run(self.info.ref, spec.ref, state.ptr, out.ptr, dt, 1, C.byref(res), st)
            assert res.value == out.ptr
        else:
            self.lib.rhs_scaled(self.info.ref, spec.ref, state.ptr, out.ptr, 1.0 if wrap == "rate" else dt, st)

    def apply_stage(self, state, k_out, dt, t, kind, y, ks, coefs, c_new, out2, err=None) -> bool:
        self.apply(state, k_out, "scaled", dt, t)
        return False   # the caller combines with the pointwise kernels

    def euler2(self, state, out, dt: float) -> bool:
        return False   # the second level would need the faces at t + dt


class RhsPlanningMixin:
    """``make_rhs_spec`` / ``make_pde_rhs`` / ``make_expression_rhs`` of :class:`~pde_hip.backend.HipBackendMixin`."""

    def make_rhs_spec(self, eq, state) -> RhsSpec:
        """Map a PDE object onto one of the fused device right-hand sides."""
        from .bc_expr import convert_bcs_with_expressions as _faces

        name = pde_kind(eq)
        grid = state.grid
        info = self.grid_info(grid, state.dtype)
        if state.__class__.__name__ != "ScalarField":
            msg = "hip backend steppers support a single ScalarField state"
            raise NotImplementedError(msg)
        if name == "DiffusionPDE":
            bcs = grid.get_boundary_conditions(eq.bc, rank=0)
            return RhsSpec(_abi.RHS_DIFFUSION, eq.diffusivity, info, _faces(bcs))
        if name == "CahnHilliardPDE":
            bc_c = grid.get_boundary_conditions(eq.bc_c, rank=0)
            bc_mu = grid.get_boundary_conditions(eq.bc_mu, rank=0)
            return RhsSpec(_abi.RHS_CAHN_HILLIARD, eq.interface_width, info, _faces(bc_c), _faces(bc_mu))
        if name == "PDE":
            rhs = dict(eq.rhs)
            if len(rhs) != 1:
                msg = "hip backend supports expression PDEs of a single scalar variable"
                raise NotImplementedError(msg)
            (var,) = rhs
            expr = pde_expression(eq, var)
            match = _match_expression_rhs(expr, var, dict(getattr(eq, "consts", {}) or {}))
            if match is None:
                msg = f"hip backend has no fused kernel for the expression `{expr}`"
                raise NotImplementedError(msg)
            # ONE condition per operator name (pde/pdes/pde.py:329-343): the inner and the outer laplace of the
            # Cahn-Hilliard form both use it
            bcs = grid.get_boundary_conditions(pde_bc_for(eq, var, "laplace"), rank=0)
            kind, param = match
            table = _faces(bcs)
            return RhsSpec(kind, param, info, table, _faces(bcs) if kind == _abi.RHS_CAHN_HILLIARD else None)
        msg = f"hip backend has no fused right-hand side for {name}"
        raise NotImplementedError(msg)

    def make_pde_rhs(self, eq, state):
        """``rhs(state_native, t) -> rate_native`` (base.py:634-651).

        ``state_native`` is a :class:`DeviceArray`; host valid data (what ``numpy_to_native`` passes through when it is
        called without a grid, e.g. by ``ScipySolver``) is uploaded here, where the grid is known."""
        grid = state.grid
        is_complex = np.dtype(state.dtype).kind == "c"
        info = self.grid_info(grid, real_dtype_of(state.dtype))
        comp_shape = tuple(np.shape(state.data))[: np.ndim(state.data) - len(info.shape)]   # (n,) for a FieldCollection
        if is_complex:
            comp_shape += (2,)       # planar (re, im) pairs

        def to_device(state_data):
            if isinstance(state_data, DeviceArray):
                return state_data
            if is_complex:
                return DeviceArray(info, comp_shape, complex_pairs=True).set_valid(np.asarray(state_data), self.stream)
            return DeviceArray(info, comp_shape).set_valid(np.asarray(state_data, dtype=info.dtype), self.stream)

        try:
            if is_complex:
                msg = "complex state"
                raise NotImplementedError(msg)
            spec = self.make_rhs_spec(eq, state)
        except NotImplementedError:
            erhs = self.make_expression_rhs(eq, state)   # raises NotImplementedError itself if unsupported

            def expr_rhs(state_data, t: float = 0) -> DeviceArray:
                state_data = to_device(state_data)
                out = state_data.empty_like()
                erhs.apply(state_data, out, "rate", 0.0, float(t))
                return out

            expr_rhs.expression = erhs  # type: ignore[attr-defined]
            return expr_rhs
        lib = self._lib

        def pde_rhs(state_data, t: float = 0) -> DeviceArray:
            state_data = to_device(state_data)
            out = state_data.empty_like()
            spec.update(float(t), self.stream)
            # 1.0 * (D * lap) == D * lap exactly
            lib.rhs_scaled(spec.info.ref, spec.ref, state_data.ptr, out.ptr, 1.0, self.stream)
            return out

        pde_rhs.spec = spec  # type: ignore[attr-defined]
        return pde_rhs

    # what `make_expression_rhs` needs to know about WHERE the expression is evaluated; the slab / block steppers
    # (pde_hip/distributed.py: DecomposedExpressionStepper) answer for the box of one rank
    def _expression_info(self, grid, dtype):
        return self.grid_info(grid, dtype)

    def _expression_faces(self, grid, bc, comp, part=None):
        """Face table of one operator: scalar conditions (``comp`` None), or those of component ``comp`` (k / (i, j)) of a vector /
        tensor operand; ``part`` "re" / "im": the conditions of the real / imaginary part of a complex operand."""
        from .bc_expr import convert_bcs_with_expressions, expression_faces

        if part is not None and comp is None:
            # (expression conditions of a complex field: the parts of `A + B * value` with a real `B`, pde_hip/bc_expr.py)
            return convert_bcs_with_expressions(grid.get_boundary_conditions(bc, rank=0), part=part)
        if part is not None:
            # one component of the complex vector a vector operator is applied to (`divergence(... gradient(c))` of a complex field)
            rank = 2 if isinstance(comp, tuple) else 1
            return convert_bcs(grid.get_boundary_conditions(bc, rank=rank), (grid.num_axes,) * rank, component=comp, part=part)
        if comp is None:
            bcs = grid.get_boundary_conditions(bc, rank=0)
            if not hasattr(bcs, "__iter__") and callable(getattr(bcs, "_setter", None)):
                return HostSetterTable(self, bcs, grid)      # a user function that writes the ghost cells (BoundariesSetter)
            return convert_bcs_with_expressions(bcs)
        rank = 2 if isinstance(comp, tuple) else 1
        return convert_bcs(grid.get_boundary_conditions(bc, rank=rank), (grid.num_axes,) * rank, component=comp)

    def _expression_aux(self, info, host):
        """Device copy of an array on the grid (array-valued constant, cell coordinates)."""
        return DeviceArray(info).set_valid(host, self.stream)

    def make_expression_rhs(self, eq, state):
        """Generic expression PDE (pde/pdes/pde.py) -> run-time specialised kernels (pde_hip/expr.py)."""
        from .expr import ExpressionPlan, ExpressionRhs

        builtin = class_expressions(eq) if pde_kind(eq) != "PDE" else None
        if pde_kind(eq) != "PDE" and builtin is None:
            msg = f"hip backend has no right-hand side for {eq.__class__.__name__}"
            raise NotImplementedError(msg)
        rhs = dict(builtin[0]) if builtin else dict(eq.rhs)
        variables = list(rhs)
        grid = state.grid
        is_complex = np.dtype(state.dtype).kind == "c"
        info = self._expression_info(grid, real_dtype_of(state.dtype))
        consts = dict(builtin[1]) if builtin else dict(getattr(eq, "consts", {}) or {})
        aliases = builtin[3] if builtin else {}
        if not is_complex and any(np.iscomplexobj(v) for v in consts.values()) or (not is_complex and bool(getattr(eq, "complex_valued", False))):
            msg = "hip backend: a complex-valued equation needs a complex state (py-pde's controller converts it, pde/solvers/controller.py:430-432)"
            raise NotImplementedError(msg)
        kind = state.__class__.__name__
        fields = list(state) if kind == "FieldCollection" else [state]
        kinds = [f.__class__.__name__ for f in fields]
        if len(fields) != len(variables) or any(k not in ("ScalarField", "VectorField", "Tensor2Field") for k in kinds):
            msg = "hip backend expression kernels support scalar, vector and rank-2 tensor fields (or a FieldCollection of them), one per equation"
            raise NotImplementedError(msg)
        # the state as a list of scalar components: a vector field `u` contributes `u#0`, `u#1`, ... (FieldCollection.data and
        # VectorField.data both carry the components along the first axis, pde/fields/collection.py, datafield_base.py:95)
        dim = grid.num_axes
        # a rank-2 field `S` contributes `S#0#0`, `S#0#1`, ... in C order, like `Tensor2Field.data` (dim, dim, *grid)
        flat: list[tuple[str, str, Any]] = []     # (flat name, variable, component: None / k / (i, j))
        vectors: dict[str, tuple[str, ...]] = {}
        tensors: dict[str, tuple[tuple[str, ...], ...]] = {}
        # complex fields: the real system of the parts (pde_hip/complex_expr.py); every field of the state is complex then, a scalar
        # field `u` contributes `u_re_`, `u_im_` - the planar pair of `DeviceArray(complex_pairs=True)`
        part_exprs: dict[str, str] = {}
        if is_complex:
            from .complex_expr import part_names, split_expression

            if any(k != "ScalarField" for k in kinds):
                msg = "hip backend: complex states are scalar fields (or collections of scalar fields)"
                raise NotImplementedError(msg)
            if getattr(eq, "user_funcs", None):
                msg = "hip backend: user functions in complex-valued expressions are not supported"
                raise NotImplementedError(msg)
            real_consts: dict[str, Any] = {}
            for var in variables:
                expr_src = rhs[var] if builtin else pde_expression(eq, var)
                def coupled(base, var=var):
                    # (conditions with complex factors couple the parts: complex_expr.COUPLING_SUFFIXES)
                    from .faces import has_complex_factors

                    try:
                        bc = builtin[2][(var, base)] if builtin else pde_bc_for(eq, var, base)
                        return has_complex_factors(grid.get_boundary_conditions(bc, rank=0))
                    except (KeyError, NotImplementedError):
                        return False

                re_s, im_s, keep, more = split_expression(str(expr_src), variables, consts, tuple(grid.axes), aliases, coupled=coupled)
                real_consts.update(keep)
                aliases = {**aliases, **more}
                part_exprs[part_names(var)[0]], part_exprs[part_names(var)[1]] = re_s, im_s
            consts = real_consts
        for var, k in zip(variables, kinds):
            if is_complex:
                flat += [(part_names(var)[0], var, "re"), (part_names(var)[1], var, "im")]
            elif k == "VectorField":
                vectors[var] = tuple(f"{var}#{c}" for c in range(dim))
                flat += [(f"{var}#{c}", var, c) for c in range(dim)]
            elif k == "Tensor2Field":
                tensors[var] = tuple(tuple(f"{var}#{i}#{j}" for j in range(dim)) for i in range(dim))
                flat += [(f"{var}#{i}#{j}", var, (i, j)) for i in range(dim) for j in range(dim)]
            else:
                flat.append((var, var, None))
        if builtin and (vectors or tensors):
            msg = f"hip backend: {eq.__class__.__name__} takes scalar fields"
            raise NotImplementedError(msg)

        def tables_for(var, plan, part=None):
            # one face table per operator NAME in the equation of `var`, like the reference (pde/pdes/pde.py:329-343)
            # (`part` not None: the equation of one part of a complex field; its operators are tagged with the part of THEIR operand)
            tables: dict[str, Any] = {}
            specs: list[tuple[Any, Any, Any]] = []
            for op in plan.operators_used:
                # components of the vector operators take the conditions of `gradient` (scalar argument) resp. of component k
                # of the vector that `divergence` is applied to (rank-1 conditions)
                base, comp, op_part = op, None, None
                if part is not None:
                    # complex fields: the operand of `<op>_imop` is the imaginary part of the operator's complex argument (complex_expr.py)
                    from .complex_expr import IM_OPERAND

                    from .complex_expr import COUPLING_SUFFIXES

                    op_part = "im" if op.endswith(IM_OPERAND) else "re"
                    base = base[: -len(IM_OPERAND)] if base.endswith(IM_OPERAND) else base
                    for sfx, sfx_part in COUPLING_SUFFIXES.items():
                        if op.endswith(sfx):
                            op_part, base = sfx_part, op[: -len(sfx)]
                    if base.startswith("gradient_squared_d"):   # the central differences inside gradient_squared of a complex argument
                        base = "gradient_squared"
                if op in getattr(plan, "vector_ops", {}):
                    idx = [int(x) for x in base.split("_")[1:]]
                    base, comp = {"grad": ("gradient", None), "div": ("divergence", idx[0]), "vlap": ("vector_laplace", idx[0]),
                                  "vgrad": ("vector_gradient", idx[0]), "tdiv": ("tensor_divergence", tuple(idx))}[base.split("_")[0]]
                bc = builtin[2][(var, base)] if builtin else pde_bc_for(eq, var, base)
                key = (comp, op_part)
                for other, other_key, table in specs:   # equal conditions share one table object (ExpressionRhs compares identities)
                    try:
                        same = other_key == key and (other is bc or bool(other == bc))
                    except (ValueError, TypeError):   # array-valued entries do not compare to a bool
                        same = False
                    if same:
                        tables[op] = table
                        break
                else:
                    tables[op] = self._expression_faces(grid, bc, comp, op_part)
                    specs.append((bc, key, tables[op]))
            return tables

        # further arrays an expression may name: array-valued constants (fields or arrays on the grid, pde/pdes/pde.py:170-185)
        # and the cell coordinates of position-dependent expressions (pde/pdes/pde.py:441-447); uploaded once, on first use
        nd = grid.num_axes
        aux_host: dict[str, Any] = {}
        for k, v in list(consts.items()):
            if np.isscalar(v):
                continue
            arr = np.asarray(getattr(v, "data", v))
            if arr.shape != tuple(grid.shape) or np.iscomplexobj(arr):
                msg = f"hip backend: constant `{k}` must be a number or a real scalar field / array on the grid"
                raise NotImplementedError(msg)
            aux_host[k] = arr
        for i, ax in enumerate(grid.axes):
            if ax not in consts and ax not in variables:
                aux_host.setdefault(ax, (lambda i=i: np.ascontiguousarray(grid.cell_coords[..., i])))
        aux_dev: dict[str, DeviceArray] = {}

        def aux_for(plan):
            for name in plan.aux_used:
                if name not in aux_dev:
                    host = aux_host[name]
                    host = host() if callable(host) else host
                    aux_dev[name] = self._expression_aux(info, np.asarray(host, dtype=info.dtype))
            return {name: aux_dev[name] for name in plan.aux_used}

        # Python functions the expressions may call (`user_funcs` of pde.PDE, pde/pdes/pde.py:84): traced symbolically by the plan
        user_funcs: dict[str, Any] = dict(getattr(eq, "user_funcs", None) or {})
        for e in (getattr(eq, "_rhs_expr", None) or {}).values():
            user_funcs.update(getattr(e, "user_funcs", None) or {})
        parts = []
        names = [name for name, _, _ in flat]
        for name, var, comp in flat:
            try:
                source = part_exprs[name] if is_complex else (rhs[var] if builtin else pde_expression(eq, var))
                plan = ExpressionPlan(source, name, consts, others=tuple(n for n in names if n != name),
                                      axes=tuple(grid.axes), aliases=aliases, aux=tuple(a for a in aux_host if a not in vectors and a not in tensors),
                                      vectors=vectors, component=None if is_complex else comp, user_funcs=user_funcs, tensors=tensors)
            except ValueError as err:
                if "unknown symbol" in str(err):   # the reference's error for this case (pde/pdes/pde.py:455-459)
                    msg = f"Undefined variable in expression for rhs of `{var}`: {err}"
                    raise RuntimeError(msg) from err
                raise
            parts.append(ExpressionRhs(self, plan, info, tables_for(var, plan, comp if is_complex else None), aux_for(plan)))
        variables = names
        if len(parts) == 1:
            return parts[0]
        from .expr import SystemRhs

        system = SystemRhs(variables, parts, info)
        system.complex_pairs = is_complex          # the state is complex: planar (re, im) pairs, modulus error norm
        return system
