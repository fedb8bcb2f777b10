"""The ``hip`` backend: py-pde's backend plugin surface implemented on libpdehip.so.

:class:`HipBackendMixin` implements the methods of ``pde.backends.base.BackendBase``
(``pde/backends/base.py:65-755``) that the finite-difference / explicit-stepper hot path uses:
``register_operator`` / ``get_operator_info`` (``:256-376``), ``make_operator_no_bc``
(``:482-521``), ``make_operator`` (``:523-565``), ``make_ghost_cell_setter`` /
``make_valid_data_setter`` / ``make_full_data_setter`` (``:378-429``), ``_apply_operator``
(``:239-254``), ``numpy_to_native`` / ``native_to_numpy`` (``:186-205``), ``make_pde_rhs``
(``:634-651``) and ``make_stepper`` (``:728-755``).

The mixin only duck-types grids, boundary conditions, PDEs and solvers, so the same code
serves (a) the stand-alone mirror classes of this package and (b) real py-pde objects when the
backend is registered as a py-pde plugin (``pde_hip/pypde_plugin.py``).

Native arrays are :class:`~pde_hip.device.DeviceArray` (ghost padded, device resident).
There is no CPU path: everything below ends in a libpdehip call or raises.
"""

from __future__ import annotations

import ctypes as C
import os
import inspect
import logging
from collections import defaultdict
from typing import Any, Callable, NamedTuple

import numpy as np

from . import _abi
from ._lib import require_device
from .device import DeviceArray, DeviceBuffer, DeviceScalar, GridInfo, ptr_array

_logger = logging.getLogger("pde_hip.backend")


# operators that are not linear in their argument: real and imaginary part of a complex field cannot go through them separately
_NONLINEAR_OPERATORS = frozenset({"gradient_squared"})


class OperatorInfo(NamedTuple):
    """Stores information about an operator (same fields as ``pde.grids.base.OperatorInfo``)."""

    factory: Callable
    rank_in: int
    rank_out: int
    name: str = ""


_FASTMATH_APPLIED: bool | None = None   # the arithmetic mode last handed to the library (HipBackendMixin._lib)


# ---------------------------------------------------------------------------------------------
# boundary conditions -> pdehip_bc_face_t[6]
# ---------------------------------------------------------------------------------------------
class FaceTable:
    """ctypes face table + the device arrays it points to (kept alive with it)."""

    def __init__(self):
        self.c = _abi.FaceArray()
        self.keepalive: list[DeviceBuffer] = []
        # the ctypes array itself also references the buffers, so `convert_bcs(...).c` is safe to
        # pass on after the FaceTable object went out of scope
        self.c._keepalive = self.keepalive
        for i in range(2 * _abi.MAX_DIM):
            self.c[i].kind = _abi.BC_SKIP

    def copy_into(self, dst) -> None:
        for i in range(2 * _abi.MAX_DIM):
            dst[i] = self.c[i]


def _upload_f64(arr: np.ndarray) -> DeviceBuffer:
    arr = np.ascontiguousarray(arr, dtype=np.float64)
    buf = DeviceBuffer(arr.nbytes)
    require_device().memcpy_h2d(buf.ptr, arr.ctypes.data, arr.nbytes, None)
    return buf


def real_dtype_of(dtype) -> np.dtype:
    """The real type that carries ``dtype`` on the device (complex data: planar real and imaginary part, pde_hip/complex_expr.py)."""
    dt = np.dtype(dtype if dtype is not None else np.float64)
    if dt.kind == "c":
        return np.dtype(np.float64 if dt == np.complex128 else np.float32)
    return dt


def convert_bcs(bcs, comp_shape: tuple[int, ...] = (), *, skip: set[tuple[int, bool]] | None = None, upload=None,
                component: int | tuple[int, ...] | None = None, part: str | None = None) -> FaceTable:
    """Reduce a ``BoundariesList`` (mirror or real py-pde) to the C face table.

    ``skip`` lists (axis, upper) faces that are filled by a halo exchange instead.  ``upload``
    turns an fp64 host array into an object with a ``.ptr`` (default: copy to the device).
    ``component``: the table of ONE component of a vector field's conditions (``comp_shape == (dim,)``) as a table for a
    scalar array - the terms of ``divergence`` inside expression PDEs are evaluated component by component.
    ``part``: "re" / "im" - the table for the real / imaginary part of a COMPLEX field: the virtual point ``const + factor * value``
    splits into the parts as long as the factors are real (value, derivative and curvature conditions with complex values; a mixed
    condition with a complex coefficient would couple the parts and is refused).
    """
    if upload is None:
        upload = _upload_f64
    if not hasattr(bcs, "__iter__"):
        # BoundariesSetter & co: opaque python callables (pde/grids/boundaries/axes.py:504)
        msg = "hip backend needs a BoundariesList of constant conditions"
        raise NotImplementedError(msg)
    grid = bcs.grid
    table = FaceTable()
    for ax, bc_axis in enumerate(bcs):
        for upper, bc in ((False, bc_axis.low), (True, bc_axis.high)):
            face = table.c[2 * ax + int(upper)]
            if skip and (ax, upper) in skip:
                continue
            get = getattr(bc, "get_virtual_point_data", None)
            if get is None or type(bc).__name__ in {"ExpressionBC", "ExpressionValueBC", "ExpressionDerivativeBC", "ExpressionMixedBC", "UserBC", "_MPIBC"}:
                msg = f"hip backend does not support boundary condition {type(bc).__name__} (needs run-time code generation)"
                raise NotImplementedError(msg)
            data = get()
            if len(data) == 3:
                const, f1, i1 = data
                f2, i2, kind = 0.0, 0, _abi.BC_ORDER1
            elif len(data) == 5:
                const, f1, i1, f2, i2 = data
                kind = _abi.BC_ORDER2
            else:
                msg = f"unexpected virtual point data of {type(bc).__name__}"
                raise NotImplementedError(msg)
            face.kind = kind
            face.index1, face.index2 = int(i1), int(i2)
            normal = bool(getattr(bc, "normal", False))
            if normal and component is not None:
                msg = "hip backend: `normal_*` conditions of a vector inside an expression are not supported"
                raise NotImplementedError(msg)
            face.flags = _abi.BCF_NORMAL if normal else 0
            if part is not None or any(np.iscomplexobj(v) for v in (const, f1, f2)):
                if part is None:
                    msg = "hip backend: complex-valued boundary conditions need a complex field"
                    raise NotImplementedError(msg)
                if np.any(np.imag(f1) != 0) or np.any(np.imag(f2) != 0):
                    msg = "hip backend: boundary conditions with complex coefficients of the field value couple real and imaginary part"
                    raise NotImplementedError(msg)
                const = np.real(const) if part == "re" else np.imag(const)
                f1, f2 = np.real(f1), np.real(f2)
            const, f1, f2 = np.asarray(const, dtype=np.float64), np.asarray(f1, dtype=np.float64), np.asarray(f2, dtype=np.float64)
            if const.ndim == 0 and f1.ndim == 0 and f2.ndim == 0:
                face.const_v, face.factor1, face.factor2 = float(const), float(f1), float(f2)
                continue
            # per (component, face cell) arrays.  Homogeneous tensor values have shape (dim,)*rank
            # (local.py:1341-1352) and broadcast over the face; inhomogeneous ones carry the face.
            face_shape = tuple(n for a, n in enumerate(grid.shape) if a != ax)
            # normal conditions act on the component along the axis; the ghost kernel indexes their arrays per face cell only,
            # which covers vector fields.  On tensor fields their values carry the remaining tensor axes (`_shape_tensor`,
            # pde/grids/boundaries/local.py:190-197): refused instead of being broadcast against the face
            if normal and len(comp_shape) > 1:
                msg = "hip backend: array-valued normal boundary conditions on tensor fields are not supported"
                raise NotImplementedError(msg)
            lead = () if normal else tuple(comp_shape)
            target = lead + face_shape

            def expand(v: np.ndarray) -> np.ndarray:
                if bool(getattr(bc, "homogeneous", v.ndim <= len(lead))) and v.ndim <= len(lead):
                    v = v.reshape(v.shape + (1,) * len(face_shape))
                v = np.broadcast_to(v, target)
                return v if component is None else v[component]

            face.flags |= _abi.BCF_ARRAYS
            for name, v in (("const_arr", const), ("factor1_arr", f1), ("factor2_arr", f2)):
                if name == "factor2_arr" and kind == _abi.BC_ORDER1:
                    continue
                buf = upload(np.ascontiguousarray(expand(v), dtype=np.float64))
                table.keepalive.append(buf)
                setattr(face, name, buf.ptr)
    return table


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


class HostSetterTable:
    """Boundary conditions given as a Python FUNCTION that writes the ghost cells of a full array (``BoundariesSetter``,
    pde/grids/boundaries/axes.py:504-560: ``setter(data_full, args)``).  Arbitrary numpy code cannot run on the device: before every pass
    that applies operators with these conditions the operand crosses PCIe twice (download -> user function -> upload) and the kernels
    then read the ghost cells from memory (every face SKIP).  The interface of the expression face tables (pde_hip/bc_expr.py):
    `time_dependent` + `reads_value` make every evaluation refresh from the pass's own input, `host_only` keeps the C loops away."""

    time_dependent = True
    reads_value = True
    host_only = True

    def __init__(self, backend, bcs, grid):
        self.backend, self.bcs, self.grid = backend, bcs, grid
        self.table = FaceTable()                 # all faces SKIP: ghost cells come from memory
        self.c = self.table.c
        _logger.warning("boundary conditions set by a Python function run on the host: the field crosses PCIe twice per operator application")

    def copy_into(self, dst) -> None:
        self.table.copy_into(dst)

    def update(self, args=None, state=None, stream=None) -> None:
        if state is None:
            msg = "hip backend: a ghost-cell setter function needs the field it is applied to"
            raise NotImplementedError(msg)
        host = state.get_hostfull(stream=stream)
        res = self.bcs._setter(host, args=dict(args or {}))
        state.set_hostfull(host if res is None else np.asarray(res), stream)


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


def make_face_setter(backend, bcs, comp_shape: tuple[int, ...] = ()):
    """``f(data_full: DeviceArray, args)`` setting all ghost faces of one field (constant-coefficient faces in one
    launch of the ghost kernel; expression faces — ``pde_hip/bc_expr.py`` — refresh their coefficient arrays first when
    they depend on time)."""
    from .bc_expr import convert_bcs_with_expressions

    table = convert_bcs_with_expressions(bcs, comp_shape)
    lib = backend._lib

    def set_faces(data_full: DeviceArray, args=None) -> None:
        table.update(args, state=data_full, stream=backend.stream)   # (conditions that are not affine in the adjacent value read it from `data_full`)
        lib.set_ghost_cells(data_full.info.ref, data_full.ncomp, table.c, data_full.ptr, backend.stream)

    set_faces.table = table   # type: ignore[attr-defined]
    return set_faces


class HipBackendMixin:
    """Implementation shared by the stand-alone and the py-pde-plugin backend classes."""

    implementation = "hip"
    copy_data = True
    supports_mpi = False

    # set by concrete classes: _operators (per class), name, config
    def _hip_init(self, device: int | None = None) -> None:
        """Remember the requested device.  NOTHING here touches the HIP runtime: py-pde instantiates every
        registered backend just to list operators (``pde/grids/base.py:1128-1150``) and only tolerates
        ``ImportError`` there (``pde/backends/registry.py:241-245``), so construction must succeed on a box
        without a GPU.  The first compute call selects the device and raises ``RuntimeError`` without one."""
        self._device_request = None if device is None else int(device)
        self._fastmath: bool | None = None   # None: from the configuration / PDEHIP_FASTMATH (see `fastmath`)
        self.stream = None  # HIP default stream; multi-GPU paths create their own
        self._info_cache: dict[tuple, GridInfo] = {}

    @property
    def fastmath(self) -> bool:
        """Arithmetic mode of the stencil kernels: False (default) = every rounding of the reference's expression order, bit-identical to its
        numpy / torch-CPU evaluation; True = the same kernels compiled with FMA contraction, like the reference's numba backend under its default
        ``fastmath`` (``pde/backends/numba/utils.py:330-336``, config ``backend.numba.fastmath``) - within 1e-10 of the exact build.
        ``config["backend.hip.fastmath"]`` with py-pde, ``backend.fastmath = True`` or ``PDEHIP_FASTMATH=1`` otherwise."""
        if self._fastmath is not None:
            return self._fastmath
        try:
            if "fastmath" in self.config:
                return bool(self.config["fastmath"])
        except TypeError:
            pass
        return os.environ.get("PDEHIP_FASTMATH", "0") == "1"

    @fastmath.setter
    def fastmath(self, value) -> None:
        self._fastmath = None if value is None else bool(value)

    @property
    def _lib(self):
        """libpdehip with the device of this backend selected (loud failure without library / GPU) and its arithmetic mode applied."""
        global _FASTMATH_APPLIED
        lib = require_device(self._device_request)
        want = self.fastmath
        if want is not _FASTMATH_APPLIED:
            lib.set_fastmath(1 if want else 0)     # (process-wide in the library: one mode at a time)
            _FASTMATH_APPLIED = want
        return lib

    @property
    def device(self) -> int:
        from ._lib import current_device, default_device

        if self._device_request is not None:
            return self._device_request
        cur = current_device()
        return default_device() if cur is None else cur

    @property
    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        self._lib.device_name(buf, 256)
        return buf.value.decode()

    @property
    def info(self) -> dict[str, Any]:
        try:
            device = self.device_name
        except (RuntimeError, ImportError) as err:   # diagnostics must not fail on a box without GPU
            device = f"unavailable ({err})"
        return {"name": self.name, "implementation": self.implementation, "device": device}

    # --- helpers -----------------------------------------------------------------------------
    def grid_info(self, grid, dtype) -> GridInfo:
        """Cached POD description of (grid, dtype)."""
        if dtype is None:
            dtype = np.float64
        dt = np.dtype(dtype)
        _abi.dtype_code(dt)  # raises NotImplementedError for complex (SURVEY.md §8b dtype)
        key = (tuple(grid.shape), tuple(float(d) for d in grid.discretization), dt.str)
        if key not in self._info_cache:
            if not all(hasattr(grid, a) for a in ("shape", "discretization", "periodic")) or len(grid.shape) != getattr(grid, "dim", len(grid.shape)):
                msg = f"hip backend only supports Cartesian grids (got {grid.__class__.__name__})"
                raise NotImplementedError(msg)
            self._info_cache[key] = GridInfo(grid.shape, grid.discretization, dt)
        return self._info_cache[key]

    def synchronize(self) -> None:
        self._lib.stream_synchronize(self.stream)

    # --- data movement -------------------------------------------------------------------------
    def numpy_to_native(self, value, grid=None):
        """Valid host data → :class:`DeviceArray`.  py-pde calls this without a grid (``pde/backends/base.py:186-194``;
        e.g. ``ScipySolver``, ``pde/solvers/scipy.py:77-79``); the geometry is then unknown here, so the host array is
        passed through and the native callables of this backend (operators, right-hand sides, ghost-cell setters) — which
        know their grid — move host input to the device themselves."""
        if isinstance(value, DeviceArray) or not isinstance(value, np.ndarray):
            return value
        if grid is None:
            return value
        info = self.grid_info(grid, value.dtype)
        comp_shape = value.shape[: value.ndim - len(info.shape)]
        return DeviceArray(info, comp_shape).set_valid(value, self.stream)

    def native_to_numpy(self, value):
        if isinstance(value, DeviceArray):
            return value.get_valid(stream=self.stream)
        return value

    def compile_function(self, func, **kwargs):
        return func

    # --- operator registry (pde/backends/base.py:256-376) ------------------------------------------
    @classmethod
    def register_operator(cls, grid_cls, name: str, factory_func=None, *, rank_in: int = 0, rank_out: int = 0):
        def register(factory):
            cls._operators[grid_cls][name] = OperatorInfo(factory=factory, rank_in=rank_in, rank_out=rank_out, name=name)
            return factory

        if factory_func is None:
            return register
        register(factory_func)
        return None

    def get_registered_operators(self, grid_id) -> set[str]:
        grid_cls = grid_id if inspect.isclass(grid_id) else grid_id.__class__
        ops: set[str] = set()
        for backend_cls in inspect.getmro(self.__class__)[:-1]:
            table = getattr(backend_cls, "_operators", {})
            for gcls in inspect.getmro(grid_cls)[:-1]:
                ops |= set(table.get(gcls, {}))
        return ops

    def get_operator_info(self, grid, operator):
        if not isinstance(operator, str):
            return operator
        for backend_cls in inspect.getmro(self.__class__)[:-1]:
            table = getattr(backend_cls, "_operators", {})
            for gcls in inspect.getmro(grid.__class__)[:-1]:
                if operator in table.get(gcls, {}):
                    return table[gcls][operator]
        # pattern operators `d_d<axis>[_central|_forward|_backward]` and `d2_d<axis>2`
        # (pde/backends/numba/backend.py:143-173)
        import functools

        from .operators import make_axis_derivative

        axes = list(getattr(grid, "axes", []))
        if operator.startswith("d_d"):
            axis_name, method = operator[len("d_d"):], "central"
            for direction in ("central", "forward", "backward"):
                if axis_name.endswith("_" + direction):
                    method, axis_name = direction, axis_name[: -len("_" + direction)]
                    break
            if axis_name in axes:
                factory = functools.partial(make_axis_derivative, axis=axes.index(axis_name), order=1, method=method)
                return OperatorInfo(factory, rank_in=0, rank_out=0, name=operator)
        if operator.startswith("d2_d") and operator.endswith("2") and operator[len("d2_d"):-1] in axes:
            factory = functools.partial(make_axis_derivative, axis=axes.index(operator[len("d2_d"):-1]), order=2)
            return OperatorInfo(factory, rank_in=0, rank_out=0, name=operator)
        msg = (
            f"Backend `{self.name}` does not define operator '{operator}' for grid "
            f"`{grid.__class__.__name__}`. Defined operators are: {sorted(self.get_registered_operators(grid))}."
        )
        raise NotImplementedError(msg)

    # --- ghost cells (pde/backends/base.py:378-429) -------------------------------------------------
    def make_ghost_cell_setter(self, bcs):
        """``f(data_full, args=None)`` — one fused kernel for all faces.

        ``data_full`` is a :class:`DeviceArray` (the normal case inside steppers and operators) or, like the reference's
        setters (``pde/backends/numba/backend.py:342-404``), a host full array (``field._data_full``) that is updated in
        place through a device round trip.
        """
        tables: dict[tuple, Any] = {}
        grid = bcs.grid
        nd = len(grid.shape)

        def ghost_cell_setter(data_full, args=None) -> None:
            if not isinstance(data_full, DeviceArray):
                host = data_full
                info = self.grid_info(grid, host.dtype)
                dev = DeviceArray(info, host.shape[: host.ndim - nd]).set_hostfull(host, self.stream)
                ghost_cell_setter(dev, args=args)
                host[...] = dev.get_hostfull(stream=self.stream)
                return
            key = data_full.comp_shape
            if key not in tables:
                tables[key] = make_face_setter(self, bcs, key)
            tables[key](data_full, args)

        return ghost_cell_setter

    def make_valid_data_setter(self, grid, rank: int = 0):
        nd = len(grid.shape)

        def set_valid(data_full, data_valid, args=None) -> None:
            if not isinstance(data_full, DeviceArray):
                # host full array: plain interior assignment (pde/backends/numpy/backend.py:72-115)
                data_full[(...,) + (slice(1, -1),) * nd] = np.asarray(data_valid)
            elif isinstance(data_valid, DeviceArray):
                # interior copy on the device: out = y + 0 is not bit-safe for -0.0, so copy bytes
                self._lib.memcpy_d2d(data_full.ptr, data_valid.ptr, data_full.nbytes, self.stream)
            else:
                data_full.set_valid(np.asarray(data_valid), self.stream)

        return set_valid

    def make_full_data_setter(self, bcs):
        set_valid = self.make_valid_data_setter(bcs.grid, 0)
        set_bcs = self.make_ghost_cell_setter(bcs)

        def set_valid_and_bcs(data_full, data_valid, args=None) -> None:
            set_valid(data_full, data_valid)
            set_bcs(data_full, args=args)

        return set_valid_and_bcs

    # --- reductions on the device (pde/backends/numba/backend.py:555-652) --------------------------------------------
    def make_integrator(self, grid, *, dtype=None):
        """``integrate(arr) -> float | ndarray``: integral over the grid, one value per tensor component, computed on the
        device (``pdehip_integrate``: cell volume x sum, two deterministic passes) — only ``ncomp`` doubles cross PCIe.
        ``arr`` is a :class:`DeviceArray`; host valid data is uploaded first (convenience, like the operators)."""
        nd = len(grid.shape)
        cell_volume = float(np.prod(grid.discretization))

        def integrate(arr):
            if not isinstance(arr, DeviceArray):
                host = np.asarray(arr)
                arr = DeviceArray(self.grid_info(grid, host.dtype), host.shape[: host.ndim - nd]).set_valid(host, self.stream)
            out = DeviceBuffer(8 * arr.ncomp)
            self._lib.integrate(arr.info.ref, arr.ncomp, arr.ptr, cell_volume, out.ptr, self.stream)
            host = np.empty(arr.ncomp, dtype=np.float64)
            self._lib.memcpy_d2h(host.ctypes.data, out.ptr, host.nbytes, self.stream)
            return float(host[0]) if not arr.comp_shape else host.reshape(arr.comp_shape)

        return integrate

    def make_finite_check(self, grid=None):
        """``is_finite(field_or_array) -> bool``: the check of the reference's ``ConsistencyTracker``
        (``np.all(np.isfinite(field.data))``, pde/trackers/trackers.py:974-1003) evaluated ON THE DEVICE (``pdehip_count_nonfinite``):
        for a :class:`DeviceArray`, or for a field whose state lives on the device between tracker interrupts
        (:class:`ResidentState`), 8 bytes per component cross PCIe instead of the whole state.  Host data is checked on the host."""

        def is_finite(obj) -> bool:
            arr = obj
            if not isinstance(obj, DeviceArray):
                link = getattr(obj, "__dict__", {}).get("_hip_link")
                if link is not None and link.host_stale:       # the device copy is the current one
                    arr = link.dev_state
                else:
                    return bool(np.all(np.isfinite(getattr(obj, "data", obj))))
            out = DeviceBuffer(8 * arr.ncomp)
            self._lib.count_nonfinite(arr.info.ref, arr.ncomp, arr.ptr, out.ptr, self.stream)
            host = np.empty(arr.ncomp, dtype=np.float64)
            self._lib.memcpy_d2h(host.ctypes.data, out.ptr, host.nbytes, self.stream)
            return not host.any()

        return is_finite

    # --- operators ------------------------------------------------------------------------------------
    def make_operator_no_bc(self, grid, operator, *, dtype=None, **kwargs):
        """``impl(arr_full: DeviceArray, out: DeviceArray)``; ghost cells are the caller's job."""
        info = self.get_operator_info(grid, operator)
        return info.factory(grid, backend=self, **kwargs)

    def _apply_operator(self, func, *values: np.ndarray, out: np.ndarray, grid=None, **kwargs) -> None:
        """Apply a native operator to host FULL arrays and write host ``out`` (base.py:239-254).

        ``values`` are the reference's compact full arrays (``field._data_full``); ``out`` is
        usually a strided interior view (fields/datafield_base.py:948).
        """
        if grid is None:
            grid = getattr(func, "grid", None)
        if grid is None:
            msg = "hip backend: operator does not know its grid"
            raise TypeError(msg)
        nd = len(grid.shape)
        if any(np.iscomplexobj(v) for v in values):
            # complex fields: the stencils have real coefficients - real and imaginary part separately (ghost cells are set already)
            if getattr(func, "__name__", "") in _NONLINEAR_OPERATORS:
                msg = f"hip backend: operator `{func.__name__}` on complex fields is not supported"
                raise NotImplementedError(msg)
            parts = []
            for take in (np.real, np.imag):
                natives = []
                for v in values:
                    info = self.grid_info(grid, real_dtype_of(v.dtype))
                    natives.append(DeviceArray(info, v.shape[: v.ndim - nd]).set_hostfull(np.ascontiguousarray(take(v)), self.stream))
                res = DeviceArray(natives[0].info, out.shape[: out.ndim - nd])
                func(*natives, res, **kwargs)
                parts.append(res.get_valid(stream=self.stream))
            out[...] = parts[0] + 1j * parts[1]
            return
        natives = []
        for v in values:
            info = self.grid_info(grid, v.dtype)
            natives.append(DeviceArray(info, v.shape[: v.ndim - nd]).set_hostfull(v, self.stream))
        info = natives[0].info
        res = DeviceArray(info, out.shape[: out.ndim - nd])
        func(*natives, res, **kwargs)
        res.get_valid(out=out, stream=self.stream)

    def make_operator(self, grid, operator, *, bcs, dtype=None, **kwargs):
        """``op(arr, out=None, args=None) -> out`` with BCs (base.py:523-565, numpy/backend.py:178-255).

        ``arr`` is a :class:`DeviceArray` (the ghost cells of ``arr`` itself are set in place —
        the valid data is untouched) or, for convenience, host valid data, in which case host data
        is returned.
        """
        info = self.get_operator_info(grid, operator)
        op_no_bc = info.factory(grid, backend=self, **kwargs)
        nd = len(grid.shape)
        shape_in = (grid.dim,) * info.rank_in + tuple(grid.shape)
        shape_out = (grid.dim,) * info.rank_out + tuple(grid.shape)
        if dtype is not None and np.dtype(dtype).kind == "c":
            return self._make_complex_operator(grid, operator, info, op_no_bc, bcs, dtype, shape_in, shape_out)
        set_ghosts = self.make_ghost_cell_setter(bcs)

        def apply_op(arr, out=None, args=None):
            host = not isinstance(arr, DeviceArray)
            if tuple(arr.shape) != shape_in:
                msg = f"Incompatible shapes {tuple(arr.shape)} != {shape_in}"
                raise ValueError(msg)
            if out is not None and tuple(out.shape) != shape_out:
                msg = f"Incompatible shapes {tuple(out.shape)} != {shape_out}"
                raise ValueError(msg)
            ginfo = self.grid_info(grid, arr.dtype if dtype is None or not host else dtype)
            native = DeviceArray(ginfo, shape_in[: len(shape_in) - nd]).set_valid(np.asarray(arr), self.stream) if host else arr
            set_ghosts(native, args=args)
            res = out if isinstance(out, DeviceArray) else DeviceArray(native.info, shape_out[: len(shape_out) - nd])
            op_no_bc(native, res)
            if isinstance(out, DeviceArray):
                return out
            if host:
                return res.get_valid(out=out, stream=self.stream)
            return res

        apply_op.grid = grid  # type: ignore[attr-defined]
        apply_op._hip_operator = (str(getattr(info, "name", operator)), int(info.rank_in), int(info.rank_out))  # type: ignore[attr-defined]
        return apply_op

    def _make_complex_operator(self, grid, operator, info, op_no_bc, bcs, dtype, shape_in, shape_out):
        """``make_operator`` for complex data (the reference specialises its kernels for complex arrays, numba/operators/cartesian.py;
        here: real coefficients, so real and imaginary part go through the real kernels one after the other, each with its part of the
        boundary values).  Host arrays in, host arrays out: complex data lives as planar pairs on the device only inside the steppers."""
        from .bc_expr import expression_faces

        operator = getattr(info, "name", operator)       # (py-pde hands over the OperatorInfo itself, pde/grids/base.py:1254-1261)
        if info.rank_in != 0 or operator in _NONLINEAR_OPERATORS:
            msg = f"hip backend: operator `{operator}` on complex fields is not supported"
            raise NotImplementedError(msg)
        from .bc_expr import convert_bcs_with_expressions

        real = real_dtype_of(dtype)
        ginfo = self.grid_info(grid, real)
        # (expression conditions: evaluated per part for `args["t"]` before they are applied, like for real fields)
        tables = {part: convert_bcs_with_expressions(bcs, part=part) if expression_faces(bcs) else convert_bcs(bcs, part=part) for part in ("re", "im")}
        lib, nd = self._lib, len(grid.shape)

        def apply_op(arr, out=None, args=None):
            if isinstance(arr, DeviceArray):
                msg = "hip backend: operators on complex data take host arrays"
                raise NotImplementedError(msg)
            arr = np.asarray(arr)
            if tuple(arr.shape) != shape_in:
                msg = f"Incompatible shapes {tuple(arr.shape)} != {shape_in}"
                raise ValueError(msg)
            if out is not None and tuple(out.shape) != shape_out:
                msg = f"Incompatible shapes {tuple(out.shape)} != {shape_out}"
                raise ValueError(msg)
            parts = []
            for part, take in (("re", np.real), ("im", np.imag)):
                native = DeviceArray(ginfo).set_valid(np.ascontiguousarray(take(arr), dtype=real), self.stream)
                if getattr(tables[part], "time_dependent", False):
                    tables[part].update(args, state=native, stream=self.stream)
                lib.set_ghost_cells(ginfo.ref, 1, tables[part].c, native.ptr, self.stream)
                res = DeviceArray(ginfo, shape_out[: len(shape_out) - nd])
                op_no_bc(native, res)
                parts.append(res.get_valid(stream=self.stream))
            result = parts[0] + 1j * parts[1]
            if out is not None:
                out[...] = result
                return out
            return result.astype(dtype, copy=False)

        apply_op.grid = grid  # type: ignore[attr-defined]
        return apply_op

    # --- products of tensor fields (base.py:567-610) -----------------------------------------------------------
    def _make_product(self, grid, outer: bool, conjugate: bool):
        """``prod(a, b, out=None)``: host valid arrays (real or complex) or :class:`DeviceArray` operands -> the product cell by cell on
        the device (``pdehip_field_product``); host in -> host out, device in -> device out."""
        nd, dim = len(grid.shape), grid.dim
        lib = self._lib

        def to_device(v, cplx: bool, real):
            if isinstance(v, DeviceArray):
                return v
            v = np.asarray(v)
            info = self.grid_info(grid, real)
            rank = v.ndim - nd
            if cplx:
                return DeviceArray(info, (dim,) * rank + (2,), complex_pairs=True).set_valid(v.astype(np.result_type(v.dtype, np.complex64), copy=False), self.stream)
            return DeviceArray(info, (dim,) * rank).set_valid(np.ascontiguousarray(v, dtype=real), self.stream)

        def prod(a, b, out=None):
            host = not isinstance(a, DeviceArray) and not isinstance(b, DeviceArray)
            if host:
                a, b = np.asarray(a), np.asarray(b)
                cplx = np.iscomplexobj(a) or np.iscomplexobj(b)
                real = real_dtype_of(np.result_type(a.dtype, b.dtype))
                if real.kind != "f":
                    real = np.dtype(np.float64)
                rank_a, rank_b = a.ndim - nd, b.ndim - nd
            else:
                if not (isinstance(a, DeviceArray) and isinstance(b, DeviceArray)):
                    msg = "hip backend: both operands of a product on the device or both on the host"
                    raise TypeError(msg)
                cplx = bool(getattr(a, "complex_pairs", False))
                if cplx != bool(getattr(b, "complex_pairs", False)):
                    msg = "hip backend: products of a complex and a real device array are not supported"
                    raise NotImplementedError(msg)
                real = a.dtype
                rank_a, rank_b = len(a.comp_shape) - int(cplx), len(b.comp_shape) - int(cplx)
            if outer:
                if rank_a != 1 or rank_b != 1:
                    msg = "Can only define outer product between vector fields"
                    raise TypeError(msg)
                kind, rank_out = 4, 2
            else:
                if rank_a < 1 or rank_b < 1:
                    msg = "Fields in dot product must have rank >= 1"
                    raise TypeError(msg)
                kinds = {(1, 1): (0, 0), (2, 1): (1, 1), (1, 2): (2, 1), (2, 2): (3, 2)}
                if (rank_a, rank_b) not in kinds:
                    msg = f"Unsupported ranks ({rank_a}, {rank_b})"
                    raise TypeError(msg)
                kind, rank_out = kinds[rank_a, rank_b]
            if host and a.shape[rank_a:] != b.shape[rank_b:]:
                msg = "Shapes of fields are not compatible for dot product"
                raise ValueError(msg)
            da, db = to_device(a, cplx, real), to_device(b, cplx, real)
            comp = (dim,) * rank_out + ((2,) if cplx else ())
            res = out if isinstance(out, DeviceArray) else DeviceArray(da.info, comp, complex_pairs=cplx)
            lib.field_product(da.info.ref, kind, int(cplx), int(bool(conjugate) and not outer), da.ptr, db.ptr, res.ptr, self.stream)
            if isinstance(out, DeviceArray) or not host:
                return res
            data = res.get_valid(stream=self.stream)
            if out is not None:
                out[...] = data
                return out
            return data

        prod.grid = grid  # type: ignore[attr-defined]
        prod._hip_product = (bool(outer), bool(conjugate))  # type: ignore[attr-defined]
        return prod

    def make_inner_prod_operator(self, field, *, conjugate: bool = True):
        """Dot product of two tensor fields (vector . vector, tensor . vector, vector . tensor, tensor . tensor), base.py:567-587;
        numpy twin: np.einsum per rank combination (numpy/backend.py:285-337)."""
        return self._make_product(field.grid, False, conjugate)

    def make_outer_prod_operator(self, field):
        """Outer product of two vector fields (base.py:589-605, numpy/backend.py:339-363)."""
        if field.__class__.__name__ != "VectorField":
            msg = "Can only define outer product between vector fields"
            raise TypeError(msg)
        return self._make_product(field.grid, True, False)

    # --- expressions as functions (base.py:653-676) -----------------------------------------------------------------
    def make_expression_function(self, expression, *, single_arg: bool = False, user_funcs=None):
        """``f(*values)`` evaluating a sympy expression (``pde.tools.expressions``: `ScalarExpression.get_function(backend)`, `evaluate`).

        Arguments that are arrays ON A GRID (fields of any rank, real or complex, cell coordinates) are evaluated ON THE DEVICE: the
        expression is lowered component by component (vectors and tensors as arrays of scalar expressions, numpy broadcasting for
        scalar x vector) onto scalar LEAF arrays - the components of the inputs and the results of differential operators - and every
        component of the result is ONE pointwise pass of the run-time compiled kernels (the planner of the expression PDEs,
        pde_hip/expr.py).  `user_funcs` that are operators of this backend (`grid.make_operator(..., backend="hip")`, the way
        `pde.tools.expressions.evaluate` hands them over, pde/tools/expressions.py:986-1026) run on device arrays with their boundary
        conditions, the argument of an operator is evaluated first (innermost first); `dot` / `inner` / `outer` products are expanded
        symbolically (the second operand conjugated where the reference does); other user functions are traced symbolically.
        Complex values are split into real and imaginary part (`as_real_imag`), each of which is a real pass.  Host arrays in -> host
        array out; :class:`DeviceArray` in -> :class:`DeviceArray` out.

        Expressions of NUMBERS only (and of indexed parameter vectors, `allow_indexed`) carry no field data: they are evaluated where
        the numbers are, by `sympy.lambdify` (the numpy backend's way, pde/backends/numpy/backend.py:408-470)."""
        import sympy as sp

        names = [str(v) for v in expression.vars]
        consts = dict(getattr(expression, "consts", {}) or {})
        funcs = dict(getattr(expression, "user_funcs", {}) or {})
        funcs.update(user_funcs or {})
        sym_expr = getattr(expression, "_sympy_expr", None)
        if sym_expr is None:
            sym_expr = sp.sympify(str(expression))
        is_tensor_expr = not isinstance(sym_expr, sp.Basic) or isinstance(sym_expr, (sp.Array, sp.MatrixBase, sp.ImmutableDenseNDimArray))
        has_indexed = any(True for _ in getattr(sym_expr, "atoms", lambda *a: ())(sp.Indexed)) if isinstance(sym_expr, sp.Basic) else False
        host_cache: dict[str, Any] = {}
        plan_cache: dict[tuple, Any] = {}

        def host_numbers(bound: dict[str, Any]):
            """Numbers (and parameter vectors) only: nothing to offload."""
            if "f" not in host_cache:
                try:
                    from pde.tools.expressions import SPECIAL_FUNCTIONS as special
                except ImportError:      # stand-alone use without py-pde
                    special = {"Heaviside": lambda x: np.heaviside(x, 0.5), "hypot": np.hypot}
                table = {**special, **{k: v for k, v in funcs.items() if callable(v)}}
                args = [sp.IndexedBase(n) if has_indexed and any(str(a.base) == n for a in sym_expr.atoms(sp.Indexed)) else sp.Symbol(n) for n in bound]
                host_cache["f"] = sp.lambdify(args, sym_expr, modules=[table, "numpy"])
            res = host_cache["f"](*bound.values())
            if is_tensor_expr or isinstance(res, (list, tuple)):
                return np.array(np.broadcast_arrays(*[np.asarray(r) for r in np.ravel(np.asarray(res, dtype=object))]), dtype=np.result_type(*np.ravel(np.asarray(res, dtype=object)))).reshape(np.shape(res))
            return res

        def evaluate(*values):
            if single_arg:
                (packed,) = values
                values = tuple(packed[i] for i in range(len(names)))
            if len(values) != len(names):
                msg = f"expression takes {len(names)} arguments ({names}), {len(values)} given"
                raise TypeError(msg)
            bound = dict(zip(names, values))
            bound.update({k: v for k, v in consts.items() if k not in bound})
            fields = {n: v for n, v in bound.items() if isinstance(v, DeviceArray) or (v is not None and not isinstance(v, dict) and np.ndim(getattr(v, "data", v)) > 0)}
            if not fields or has_indexed:
                return host_numbers(bound)
            return _ExpressionEvaluation(self, sym_expr, bound, fields, funcs, plan_cache).run()

        return evaluate

    # --- PDE right hand sides ---------------------------------------------------------------------------
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
                re_s, im_s, keep, more = split_expression(str(expr_src), variables, consts, tuple(grid.axes), aliases)
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

                    op_part = "im" if op.endswith(IM_OPERAND) else "re"
                    base = base[: -len(IM_OPERAND)] if base.endswith(IM_OPERAND) else base
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


class _ExpressionEvaluation:
    """One call of the function :meth:`HipBackendMixin.make_expression_function` returns, for arguments that are arrays on a grid.

    Values are numpy OBJECT arrays of sympy expressions of shape ``(dim,) * rank`` over LEAF symbols, each of which stands for a real
    scalar :class:`DeviceArray` (a component of an input, of its real / imaginary part, or of the result of an operator)."""

    def __init__(self, backend, sym_expr, bound: dict, fields: dict, funcs: dict, plan_cache: dict):
        import sympy as sp

        self.sp, self.backend, self.expr, self.bound, self.funcs, self.plan_cache = sp, backend, sym_expr, bound, funcs, plan_cache
        self.leaves: dict[Any, DeviceArray] = {}
        # the grid: from an operator of this backend among the functions, else the smallest array is a scalar field
        grid = next((getattr(f, "grid", None) for f in funcs.values() if getattr(f, "grid", None) is not None), None)
        arrays = {n: (v if isinstance(v, DeviceArray) else np.asarray(getattr(v, "data", v))) for n, v in fields.items()}
        self.on_device = all(isinstance(v, DeviceArray) for v in arrays.values())
        if grid is not None:
            shape = tuple(int(n) for n in grid.shape)
        else:
            first = min(arrays.values(), key=lambda a: len(a.shape))
            shape = tuple(first.info.shape) if isinstance(first, DeviceArray) else tuple(first.shape)
        kinds = [np.dtype(a.dtype) for a in arrays.values()]
        real = real_dtype_of(np.result_type(*kinds)) if all(k.kind in "fc" for k in kinds) else np.dtype(np.float64)
        if grid is not None:
            self.info = backend.grid_info(grid, real)
        else:
            from .device import GridInfo

            cells = shape if 1 <= len(shape) <= 3 else (int(np.prod(shape)),)
            self.info = GridInfo(cells, [1.0] * len(cells), real)
        self.grid_shape, self.dim = shape, (int(grid.dim) if grid is not None else len(shape))
        self.values: dict[str, np.ndarray] = {}
        for name, arr in arrays.items():
            self.values[name] = self._input(name, arr)

    # --- leaves -------------------------------------------------------------------------------------------------------------
    def _leaf(self, dev: DeviceArray):
        sym = self.sp.Symbol(f"_leaf{len(self.leaves)}_", real=True)
        self.leaves[sym] = dev
        return sym

    def _components(self, dev: DeviceArray, comp_shape: tuple[int, ...], is_complex: bool) -> np.ndarray:
        out = np.empty(comp_shape, dtype=object)
        for idx in np.ndindex(*comp_shape) if comp_shape else [()]:
            view = dev
            for i in idx:
                view = view.component(i)
            out[idx] = (self._leaf(view.component(0)) + self.sp.I * self._leaf(view.component(1))) if is_complex else self._leaf(view)
        return out

    def _input(self, name: str, arr) -> np.ndarray:
        nd = len(self.grid_shape)
        if isinstance(arr, DeviceArray):
            cplx = bool(arr.complex_pairs)
            comp_shape = arr.comp_shape[:-1] if cplx else arr.comp_shape
            return self._components(arr, tuple(comp_shape), cplx)
        if arr.ndim < nd or tuple(arr.shape[arr.ndim - nd:]) != self.grid_shape:
            msg = f"hip backend: argument `{name}` of shape {arr.shape} does not live on the grid {self.grid_shape}"
            raise ValueError(msg)
        comp_shape = tuple(arr.shape[: arr.ndim - nd])
        cplx = np.iscomplexobj(arr)
        dev = DeviceArray(self.info, comp_shape + ((2,) if cplx else ()), complex_pairs=cplx)
        host = np.reshape(arr, comp_shape + tuple(self.info.shape))
        dev.set_valid(host if cplx else np.ascontiguousarray(host, dtype=self.info.dtype), self.backend.stream)
        return self._components(dev, comp_shape, cplx)

    # --- lowering -----------------------------------------------------------------------------------------------------------
    def lower(self, e) -> Any:
        sp = self.sp
        if isinstance(e, sp.Symbol):
            name = e.name
            if name in self.values:
                return self.values[name]
            if name in self.bound:
                v = self.bound[name]
                if v is None or isinstance(v, dict):
                    return None                      # the `none` / `bc_args` of operator signatures (pde/tools/expressions.py:1033-1036)
                return np.array(sp.sympify(complex(v) if np.iscomplexobj(v) else float(v)), dtype=object)
            msg = f"Undefined variable in expression: {name}"
            raise RuntimeError(msg)
        if isinstance(e, sp.core.function.AppliedUndef):
            return self._call(e.func.__name__, [self.lower(a) for a in e.args])
        if not e.args:
            return np.array(e, dtype=object)
        args = [self.lower(a) for a in e.args]
        if all(a.shape == () for a in args):
            return np.array(e.func(*[a.item() for a in args]), dtype=object)
        if e.is_Add:
            if len({a.shape for a in args}) != 1:
                msg = "cannot add fields of different rank"
                raise ValueError(msg)
            return np.sum(np.stack(args), axis=0)
        if e.is_Mul and sum(a.shape != () for a in args) == 1:
            out = args[0]
            for a in args[1:]:
                out = out * a
            return out
        msg = f"hip backend: expression `{e}` of vector / tensor arguments is not supported (sums, scalar multiples, products, operators)"
        raise NotImplementedError(msg)

    def _call(self, name: str, args: list) -> np.ndarray:
        sp = self.sp
        func = self.funcs.get(name)
        real_args = [a for a in args if a is not None]
        if getattr(func, "_hip_operator", None) is not None:
            _, rank_in, rank_out = func._hip_operator
            (arg,) = real_args
            if arg.ndim != rank_in:
                msg = f"operator `{name}` takes a field of rank {rank_in}"
                raise ValueError(msg)
            parts = self._split(arg)
            res = None
            for k, part in enumerate(parts):     # linear with real coefficients: real and imaginary part separately
                if k == 1 and all(x == 0 for x in part.flat):
                    continue
                out = func(self.materialise(part))
                comps = self._components(out, tuple(out.comp_shape), False)
                res = comps if res is None else res + sp.I * comps
            return res
        if getattr(func, "_hip_product", None) is not None:
            outer, conj = func._hip_product
            a, b = real_args
            if conj:
                b = np.vectorize(sp.conjugate, otypes=[object])(b)
            if outer:
                return np.multiply.outer(a, b)
            if a.ndim < 1 or b.ndim < 1:
                msg = "Fields in dot product must have rank >= 1"
                raise TypeError(msg)
            return np.tensordot(a, b, axes=(a.ndim - 1, 0))
        if callable(func):
            # a Python function of the user: traced with symbolic arguments (scalars as sympy expressions, vectors / tensors as object arrays)
            try:
                res = func(*[a.item() if a.shape == () else a for a in real_args])
            except Exception as err:   # noqa: BLE001 - whatever the user's code raises on symbolic input
                msg = f"hip backend: user function `{name}` cannot be traced symbolically ({type(err).__name__}: {err})"
                raise NotImplementedError(msg) from err
            return np.array(res, dtype=object)
        if hasattr(sp, name) and all(a.shape == () for a in real_args):
            return np.array(getattr(sp, name)(*[a.item() for a in real_args]), dtype=object)
        msg = f"hip backend: unknown function `{name}` in expression"
        raise NotImplementedError(msg)

    def _split(self, val: np.ndarray) -> list[np.ndarray]:
        """(real parts, imaginary parts) of a value; expressions without `I` of real leaves are real as they stand (`a**b` of real arrays
        is real arithmetic in numpy too - sympy would not commit itself)."""
        sp = self.sp
        if not any(sp.sympify(x).has(sp.I) for x in val.flat):
            zeros = np.empty(val.shape, dtype=object)
            zeros[...] = sp.Integer(0)
            return [val, zeros]
        parts = [np.vectorize(lambda x, k=k: sp.expand(x).as_real_imag()[k], otypes=[object])(val) for k in (0, 1)]
        for part in parts:
            for x in part.flat:
                if sp.sympify(x).atoms(sp.re, sp.im, sp.arg):
                    msg = f"hip backend: cannot split `{x}` into real and imaginary part"
                    raise NotImplementedError(msg)
        return parts

    # --- evaluation ---------------------------------------------------------------------------------------------------------
    def pointwise(self, expr) -> Any:
        """A REAL scalar expression of leaves -> a scalar :class:`DeviceArray` (or a float when no leaf is left in it)."""
        from .expr import ExpressionPlan, ExpressionRhs

        sp = self.sp
        expr = sp.sympify(expr)
        used = [s for s in self.leaves if s in expr.free_symbols]
        if not used:
            return float(expr)
        # canonical names by order of appearance: equal expressions of other leaves share one compiled pass
        renamed = {s: sp.Symbol(f"_a{k}_", real=True) for k, s in enumerate(used)}
        text = sp.sstr(expr.xreplace(renamed))
        first, others = "_a0_", tuple(f"_a{k}_" for k in range(1, len(used)))
        key = (text, self.info.key())
        if key not in self.plan_cache:
            plan = ExpressionPlan(text, first, {}, aux=others)
            self.plan_cache[key] = (plan, ExpressionRhs(self.backend, plan, self.info, {}, {n: DeviceArray(self.info) for n in others if n in plan.aux_used}))
        plan, erhs = self.plan_cache[key]
        for k, s in enumerate(used[1:], start=1):
            if f"_a{k}_" in plan.aux_used:
                erhs.aux[f"aux:_a{k}_"] = self.leaves[s]
        out = DeviceArray(self.info)
        erhs.apply(self.leaves[used[0]], out, "rate", 0.0, 0.0)
        return out

    def materialise(self, val: np.ndarray) -> DeviceArray:
        """A (real) value as ONE device array with its tensor axes: the operand of an operator."""
        lib = self.backend._lib
        dev = DeviceArray(self.info, tuple(val.shape))
        for idx in np.ndindex(*val.shape) if val.shape else [()]:
            view = dev
            for i in idx:
                view = view.component(i)
            res = self.pointwise(val[idx])
            if isinstance(res, DeviceArray):
                lib.memcpy_d2d(view.ptr, res.ptr, view.info.comp_elems * view.itemsize, self.backend.stream)
            else:
                view.set_valid(np.full(self.info.shape, res, dtype=self.info.dtype), self.backend.stream)
        return dev

    def run(self):
        sp = self.sp
        if isinstance(self.expr, sp.Basic) and not isinstance(self.expr, (sp.Array, sp.MatrixBase)):
            val = self.lower(self.expr)
        else:     # a tensor expression: its entries are scalar expressions
            entries = np.array(self.expr.tolist() if hasattr(self.expr, "tolist") else self.expr, dtype=object)
            val = np.empty(entries.shape, dtype=object)
            for idx in np.ndindex(*entries.shape):
                item = self.lower(sp.sympify(entries[idx]))
                if item.shape != ():
                    msg = "hip backend: entries of a tensor expression must be scalars"
                    raise NotImplementedError(msg)
                val[idx] = item.item()
        parts = self._split(val)
        is_complex = any(x != 0 for x in parts[1].flat)
        if self.on_device:
            if is_complex:
                msg = "hip backend: complex results of expression functions come back as host arrays"
                raise NotImplementedError(msg)
            return self.materialise(parts[0]) if val.shape else self._scalar_device(parts[0].item())
        results = []
        for part in parts[: 2 if is_complex else 1]:
            host = np.empty(tuple(val.shape) + self.grid_shape, dtype=self.info.dtype)
            numbers_only = True
            for idx in np.ndindex(*val.shape) if val.shape else [()]:
                res = self.pointwise(part[idx])
                if isinstance(res, DeviceArray):
                    numbers_only = False
                    host[idx] = res.get_valid(stream=self.backend.stream).reshape(self.grid_shape)
                else:
                    host[idx] = res
            results.append((host, numbers_only))
        if all(n for _, n in results) and not val.shape:
            # no field entered the result (`evaluate("sin", ..., consts={"sin": 3.14})`): a number, broadcast by the caller
            value = complex(results[0][0].flat[0], results[1][0].flat[0]) if is_complex else float(results[0][0].flat[0])
            return value
        return results[0][0] + 1j * results[1][0] if is_complex else results[0][0]

    def _scalar_device(self, expr) -> DeviceArray:
        res = self.pointwise(expr)
        if isinstance(res, DeviceArray):
            return res
        return DeviceArray(self.info).set_valid(np.full(self.info.shape, res, dtype=self.info.dtype), self.backend.stream)


def _config_get(config, key: str, default):
    try:
        if config is not None and key in config:
            return config[key]
    except TypeError:
        pass
    return default


_DATA_ATTRIBUTES = frozenset({"data", "_data_valid", "_data_full", "_FieldBase__data_full"})
_SYNCED_CLASSES: dict[type, type] = {}


class ResidentState:
    """Link between a host field object and its device-resident copy (no full-field PCIe traffic per
    tracker interrupt; reference behaviour being replaced: ``pde/backends/torch/backend.py:654-662``).

    The field object handed to the stepper keeps its identity (the controller returns it, trackers receive it), but its
    class is swapped for a dynamic subclass whose data attributes (``data``, ``_data_full`` ...) first bring the host
    arrays up to date — one pinned-speed download — and then count as a possible modification, so the next stepper call
    uploads again.  Nothing else about the field changes; copies of it are ordinary fields.

    Limitation: synchronisation happens on ATTRIBUTE ACCESS.  A numpy view obtained earlier (``arr = state.data``
    kept by a tracker or by the caller) is not refreshed behind the holder's back while the run goes on — it shows the state of
    its last ``state.data`` access — and writes made through such a held view after that access are not seen.  Code that wants
    the reference's behaviour (the stepper updates the host array in place at every call) sets ``resident_state=False`` in the
    backend's configuration, which restores the upload / download per stepper call of ``pde/backends/torch/backend.py:654-662``.
    """

    def __init__(self, field, dev_state: DeviceArray, backend):
        self.dev_state, self.backend = dev_state, backend
        self.host_stale = False          # device is ahead of the host arrays
        self.host_touched = True         # host arrays may differ from the device copy (initially: never uploaded)
        self.downloads = self.uploads = 0

    @staticmethod
    def attach(field, dev_state: DeviceArray, backend) -> "ResidentState":
        link = field.__dict__.get("_hip_link")
        cls = type(field)
        base = getattr(cls, "_hip_base_class", cls)
        if base not in _SYNCED_CLASSES:
            _SYNCED_CLASSES[base] = _make_synced_class(base)
        if link is not None and link.dev_state is dev_state:
            if cls is base:              # a host access since the last call put the plain class back (before_host_access)
                field.__class__ = _SYNCED_CLASSES[base]
            return link
        if link is not None:             # a stepper of an earlier run: settle it first
            link.pull(field)
        link = ResidentState(field, dev_state, backend)
        field.__dict__["_hip_link"] = link
        link._field_ref = field
        if cls is base:
            field.__class__ = _SYNCED_CLASSES[base]
        return link

    def __reduce__(self):
        # a field that was read after the run is a plain py-pde object again but still carries this link in its `__dict__` (the next
        # stepper call picks it up): copies and pickles of the field get `None` in its place
        return (type(None), ())

    def __deepcopy__(self, memo):
        return None

    def _host_valid(self):
        field = self._field_ref
        base = getattr(type(field), "_hip_base_class", type(field))
        return base.data.fget(field) if isinstance(getattr(base, "data", None), property) else object.__getattribute__(field, "data")

    def push(self) -> None:
        if self.host_touched:
            field = self._field_ref
            field.__dict__["_hip_link"] = None            # plain access while we read the host arrays
            try:
                self.dev_state.set_valid(field.data, self.backend.stream)
            finally:
                field.__dict__["_hip_link"] = self
            self.host_touched, self.host_stale = False, False
            self.uploads += 1

    def device_advanced(self) -> None:
        self.host_stale = True

    def pull(self, field=None) -> None:
        """Bring the host arrays up to date (called on the first data access after the device advanced)."""
        field = self._field_ref if field is None else field
        if self.host_stale:
            self.host_stale = False
            field.__dict__["_hip_link"] = None
            try:
                self.dev_state.get_valid(out=field.data, stream=self.backend.stream)
            finally:
                field.__dict__["_hip_link"] = self
            self.downloads += 1

    def before_host_access(self) -> None:
        """First access to the data after a stepper call: the host arrays are current from here on and may be written, so nothing
        needs intercepting until the next stepper call - the field gets its own class back (``type(result) is pde.ScalarField`` once
        the result has been looked at; `attach` swaps the intercepting subclass in again)."""
        self.pull()
        self.host_touched = True
        field = self._field_ref
        base = getattr(type(field), "_hip_base_class", None)
        if base is not None:
            object.__dict__["__class__"].__set__(field, base)


def _make_synced_class(base: type) -> type:
    def __getattribute__(self, name):
        if name in _DATA_ATTRIBUTES:
            link = object.__getattribute__(self, "__dict__").get("_hip_link")
            if link is not None:
                link.before_host_access()
        return base.__getattribute__(self, name)

    def __reduce_ex__(self, protocol):
        # pickling / deepcopy: settle the data and present the plain class
        link = self.__dict__.pop("_hip_link", None)
        if link is not None:
            self.__dict__["_hip_link"] = None
            link.pull(self)
            del self.__dict__["_hip_link"]
        self.__class__ = base
        return base.__reduce_ex__(self, protocol)

    # `field.__class__` keeps answering with the field's own class: py-pde compares classes by identity before any binary
    # operation (`assert_field_compatible`, pde/fields/base.py:385-390) and builds copies from `self.__class__`; only
    # `type(field)` shows the intercepting subclass
    real_class = object.__dict__["__class__"]

    def _get_class(self):
        return base

    def _set_class(self, value):
        real_class.__set__(self, value)

    # py-pde registers every field subclass by NAME (pde/fields/base.py:77-88) to rebuild fields from stored attributes: the
    # registry must keep pointing at the real class
    import warnings

    registry = getattr(base, "_subclasses", None)
    previous = registry.get(base.__name__) if isinstance(registry, dict) else None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        synced = type(base.__name__, (base,), {"__getattribute__": __getattribute__, "__reduce_ex__": __reduce_ex__, "_hip_base_class": base,
                                               "__class__": property(_get_class, _set_class),
                                               "__module__": base.__module__, "__qualname__": base.__qualname__, "__doc__": base.__doc__})
    if previous is not None:
        registry[base.__name__] = previous
    return synced


# ---------------------------------------------------------------------------------------------
# stand-alone backend class + registry (mirrors pde/backends/registry.py:143-230 for one name)
# ---------------------------------------------------------------------------------------------
class BackendBase:
    """Minimal stand-in for ``pde.backends.base.BackendBase`` used when py-pde is absent."""

    _operators: dict = defaultdict(dict)

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        cls._operators = defaultdict(dict)
        cls._logger = _logger.getChild(cls.__qualname__)

    def __init__(self, config=None, *, name: str | None = None):
        self.config = dict(config or {})
        self.name = name or self.__class__.__name__

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(name={self.name!r})"


class HipBackend(HipBackendMixin, BackendBase):
    """MI355X backend (``backend="hip"`` or ``"hip:<device>"``)."""

    def __init__(self, config=None, *, name: str = "hip", device: int | None = None):
        super().__init__(config, name=name)
        self._hip_init(device)

    @classmethod
    def from_args(cls, config, args: str = "", *, name: str | None = None):
        """``get_backend("hip:2")`` → device 2 (pattern of torch/backend.py:82-96)."""
        device = int(args) if args else None
        return cls(config, name=name or (f"hip:{args}" if args else "hip"), device=device)


_BACKENDS: dict[str, HipBackend] = {}


def get_backend(backend="hip") -> HipBackend:
    """Return the backend object for ``"hip"`` / ``"hip:<device>"`` or pass objects through."""
    if isinstance(backend, HipBackendMixin):
        return backend  # type: ignore[return-value]
    if not isinstance(backend, str):
        msg = f"Unknown backend {backend!r}"
        raise TypeError(msg)
    name, _, args = backend.partition(":")
    if name not in {"hip", "default", "auto"}:
        msg = f"pde_hip only provides the `hip` backend (got `{backend}`)"
        raise KeyError(msg)
    key = f"hip:{args}" if args else "hip"
    if key not in _BACKENDS:
        _BACKENDS[key] = HipBackend.from_args(None, args, name=key)
    return _BACKENDS[key]
