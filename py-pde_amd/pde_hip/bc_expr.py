"""Expression boundary conditions (``virtual_point`` / ``value_expression`` / ``derivative_expression`` /
``mixed_expression``) for the hip backend.

Reference semantics (``pde/grids/boundaries/local.py:766-1150``, numba twin
``pde/backends/numba/_boundaries.py:256-394``, torch twin ``pde/backends/torch/_boundaries.py:258-345``): the
virtual point of every face cell is ``F(value, dx, *coords, t)`` where ``value`` is the field in the adjacent cell
(or ``value_cell``), ``coords`` the wall point and ``t`` the time handed over as ``args={"t": t}``; ``F`` is
``<expr>``, ``2*(<expr>) - value``, ``dx*(<expr>) + value`` or the Robin combination (``local.py:849-866``).

Every ``F`` that is affine in ``value`` — all value / derivative conditions and mixed conditions whose coefficients do
not read the field — is exactly the constant-coefficient form the ghost kernel and the stencil kernels already evaluate,

    ghost = A(dx, coords, t) + B(dx, coords, t) * value[index]

with per-face-cell coefficient arrays (``PDEHIP_BCF_ARRAYS``, ``include/pdehip.h``).  ``A`` and ``B`` are split off
symbolically (``B = dF/dvalue``, ``A = F(value=0)``) and evaluated on the wall points once (time independent: the face
then costs nothing extra in the time loop) or whenever the time changes.  Faces that change with time are refreshed ON THE
DEVICE: ``A`` and ``B`` are printed as C, compiled at run time (``pdehip_bcprog_create``, hiprtc) into one small kernel per
face table / right-hand side that rewrites the coefficient arrays for a given ``t`` (:func:`build_program`); the C time loops
call it before every right-hand side (``pdehip_rhs_t::bc_program``), so time-dependent conditions cost one extra launch per
evaluation and no host work.  An ``F`` that is NOT affine in ``value`` (e.g. a radiation condition ``-value**4``) is the same
kind of face with ``A = F(value now, dx, coords, t)``, ``B = 0``: the program reads the adjacent cells of the field the
conditions are about to be applied to (``reads_value``), exactly when the reference evaluates ``arr[..., value_cell]``
(``local.py:1089-1135``) - before every operator application / right-hand side, Runge-Kutta stages included.  Conditions
given as Python FUNCTIONS cannot travel to the device: they are probed on the host (coefficient arrays uploaded before every
right-hand side, steps driven from Python) and must be affine in ``value``.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from . import _abi

EXPRESSION_BC_CLASSES = {"ExpressionBC", "ExpressionValueBC", "ExpressionDerivativeBC", "ExpressionMixedBC"}


def is_expression_bc(bc) -> bool:
    return any(c.__name__ in EXPRESSION_BC_CLASSES for c in type(bc).__mro__)


class _AffineFace:
    """``A`` and ``B`` of one expression face as numpy callables of ``t``.  ``window = (lo, hi)``: the box (global cell ranges
    per grid axis) of a slab / block of a decomposed grid - the face then covers the box's extent along the other axes, with the
    wall coordinates of the WHOLE grid (bit-identical to the undecomposed run; the reference rebuilds the conditions on a
    sub-grid with its own bounds, ``pde/grids/_mesh.py:535-569``), and the value cell is counted from the box's first cell."""

    def __init__(self, bc, window=None, part: str | None = None):
        """``part``: "re" / "im" - the face of the real / imaginary part of a COMPLEX field.  ``F = A + B * value`` with complex ``A``
        and REAL ``B`` (every value / derivative condition whose expression does not read the field) acts on the parts separately,
        ``ghost_re = Re A + B * value_re``, ``ghost_im = Im A + B * value_im``; a complex ``B`` or an ``F`` that is not affine in
        ``value`` couples the parts and is refused."""
        import sympy as sp

        self.part = part

        if getattr(bc, "rank", 0) != 0:
            msg = "Expression boundary conditions only work for scalar conditions"
            raise NotImplementedError(msg)
        grid = bc.grid
        self._callable = None
        self.reads_value = False
        if getattr(bc, "_is_func", False):
            # the condition is a Python function F(adjacent_value, dx, *coords, t) (pde/grids/boundaries/local.py:921-963): it
            # cannot run on the device, but its coefficient arrays can be taken on the host - before every right-hand side,
            # because nothing is known about its dependence on t - as long as it is affine in `adjacent_value`
            self._callable = bc._make_function()
            self._target = bc._input["target"]
            self._value_func = bc._prepare_function(bc._input["value_expr"])
            self.time_dependent = True
            self.needs_time = False         # a callable is evaluated at t = 0 when no time is given (local.py:1137-1146)
            self.dx = float(grid.discretization[bc.axis])
            coords = grid._boundary_coordinates(axis=bc.axis, upper=bc.upper)
            self.coords = [np.asarray(c, dtype=np.float64) for c in np.moveaxis(coords, -1, 0)]
            index = int(bc._get_value_cell_index(with_ghost_cells=False))
            self.axis, self.upper, self.grid = int(bc.axis), bool(bc.upper), grid
            self._set_window(window, index)
            self.evaluate(0.0)   # raises for functions that are not affine in the adjacent value
            return
        mirror = hasattr(bc, "virtual_point_sympy")       # pde_hip.boundaries.ExpressionBC (stand-alone mirror)
        expr = sp.sympify(bc.virtual_point_sympy if mirror else bc._func_expression._sympy_expr)
        names = ["value", "dx", *grid.axes, "t"]
        by_name = {s.name: s for s in expr.free_symbols}
        unknown = set(by_name) - set(names)
        if unknown:
            msg = f"hip backend: unknown symbol(s) {sorted(unknown)} in boundary expression `{expr}`"
            raise NotImplementedError(msg)
        value = by_name.get("value", sp.Symbol("value"))
        slope = sp.diff(expr, value)
        # not affine in `value`: A = F(value, ...), B = 0, re-evaluated from the field whenever the conditions are applied
        self.reads_value = bool(slope.has(value))
        if self.reads_value:
            offset, slope = expr, sp.Integer(0)
        else:
            offset = expr.subs(value, 0)
        if part is not None:
            if self.reads_value:
                msg = f"hip backend: the boundary expression `{expr}` of a complex field is not affine in `value` (it couples real and imaginary part)"
                raise NotImplementedError(msg)
            # dx, the coordinates and t are real: split A and B with sympy (the symbols of the parsed expression carry no assumptions)
            real = {sym: sp.Symbol(name, real=True) for name, sym in by_name.items()}
            by_name = {name: real[sym] for name, sym in by_name.items()}
            value = by_name.get("value", sp.Symbol("value", real=True))
            a_re, a_im = sp.expand(offset.xreplace(real)).as_real_imag()
            b_re, b_im = sp.expand(slope.xreplace(real)).as_real_imag()
            # (a complex slope couples the parts: the callers add the coupling terms - pde_hip/faces.py: has_complex_factors, convert_bcs -
            # from the tables "cpl-" / "cpl+" (constant 0, slope -/+ Im B) and "zero")
            if part in ("re", "im"):
                offset, slope = (a_re if part == "re" else a_im), b_re
            elif part in ("cpl-", "cpl+"):
                offset, slope = sp.Integer(0), (-b_im if part == "cpl-" else b_im)
            elif part == "zero":
                offset, slope = sp.Integer(0), sp.Integer(0)
            else:
                msg = f"unknown part `{part}`"
                raise ValueError(msg)
            if (offset.atoms(sp.re, sp.im, sp.arg) | slope.atoms(sp.re, sp.im, sp.arg)):
                msg = f"hip backend: cannot split the boundary expression `{expr}` into real and imaginary part"
                raise NotImplementedError(msg)
        args = [by_name.get(n, sp.Symbol(n)) for n in names[1:]]
        self._offset = sp.lambdify([value, *args], offset, modules="numpy")
        self._slope = sp.lambdify(args, slope, modules="numpy")
        self._symbolic = (offset, slope, {n: by_name.get(n, sp.Symbol(n)) for n in names[1:]}, value)   # for the device program
        try:
            _c_code(offset), _c_code(slope)
            self._printable = True
        except NotImplementedError:
            # no C form (a function the printer does not know): the coefficient arrays come from the host, like those of a Python
            # function - unless the condition reads the field, which only the device holds
            if self.reads_value:
                raise
            self._printable = False
        self.axis, self.upper, self.grid = int(bc.axis), bool(bc.upper), grid
        self.time_dependent = "t" in by_name or self.reads_value     # i.e. "has to be refreshed"
        self.needs_time = "t" in by_name
        self.dx = float(grid.discretization[bc.axis])
        if mirror:
            self.coords = bc.wall_coordinates()
            index = bc.value_cell_index
        else:
            coords = grid._boundary_coordinates(axis=bc.axis, upper=bc.upper)
            self.coords = [np.asarray(c, dtype=np.float64) for c in np.moveaxis(coords, -1, 0)]
            index = int(bc._get_value_cell_index(with_ghost_cells=False))
        self._set_window(window, index)

    @property
    def host_only(self) -> bool:
        """The coefficient arrays of this face are evaluated on the host (a Python function, or an expression without a C form)."""
        return self._callable is not None or not getattr(self, "_printable", True)

    def _set_window(self, window, index: int) -> None:
        """Cut the face to the box ``window`` (None: the whole grid); ``index``: value cell along the face's axis (whole grid)."""
        grid, axis = self.grid, self.axis
        nd = len(grid.shape)
        index = index if index >= 0 else index + int(grid.shape[axis])
        if window is None:
            self.lo, self.local_shape = [0] * nd, [int(n) for n in grid.shape]
        else:
            lo, hi = window
            self.lo, self.local_shape = [int(v) for v in lo], [int(h) - int(l) for l, h in zip(lo, hi)]
            if not self.lo[axis] <= index < self.lo[axis] + self.local_shape[axis]:
                msg = "boundary condition of a decomposed axis reads a cell of another slab / block"
                raise NotImplementedError(msg)
            cut = tuple(slice(self.lo[a], self.lo[a] + self.local_shape[a]) for a in range(nd) if a != axis)
            self.coords = [np.ascontiguousarray(c[cut]) for c in self.coords]
        self.face_shape = self.coords[0].shape if self.coords else ()
        self.index = index - self.lo[axis]

    def _evaluate_callable(self, t: float) -> tuple[np.ndarray, np.ndarray]:
        shape = self.face_shape
        probe = [np.full(shape, v, dtype=np.float64) for v in (0.0, 1.0, 2.0)]

        part = getattr(self, "part", None)

        def call(func, v):
            res = np.asarray(func(v, self.dx, *self.coords, t))
            if part is not None:      # complex field: the function is probed with real values; its result splits when the slope is real
                return np.array(np.broadcast_to(res.astype(np.complex128), shape), order="C")
            return np.array(np.broadcast_to(np.asarray(res, dtype=np.float64), shape), dtype=np.float64, order="C")

        def take(a, b):
            if part is None:
                return a, b
            if np.any(np.imag(b) != 0):
                msg = "hip backend: boundary condition function multiplies the field value by a complex number (it couples real and imaginary part)"
                raise NotImplementedError(msg)
            return np.ascontiguousarray(np.real(a) if part == "re" else np.imag(a)), np.ascontiguousarray(np.real(b))

        with np.errstate(all="ignore"):
            if self._target in ("value", "derivative"):
                f0, f1 = call(self._value_func, probe[0]), call(self._value_func, probe[1])
                if np.array_equal(f0, f1, equal_nan=True) and np.array_equal(f0, call(self._value_func, probe[2]), equal_nan=True):
                    # the usual case - a function of position and time only: exactly the reference's `2 f - value` / `dx f + value`
                    return take(*((2 * f0, np.full(shape, -1.0)) if self._target == "value" else (self.dx * f0, np.full(shape, 1.0))))
            a = call(self._callable, probe[0])
            b = call(self._callable, probe[1]) - a
            check = call(self._callable, probe[2])
        if not np.allclose(check, a + 2 * b, rtol=1e-12, atol=1e-12, equal_nan=True):
            msg = "hip backend: boundary condition function is not affine in the adjacent value (needs run-time code generation)"
            raise NotImplementedError(msg)
        return take(a, b)

    def evaluate(self, t: float, value: np.ndarray | None = None) -> tuple[np.ndarray, np.ndarray]:
        """``value``: the field in the value cells of the face (only read by conditions that are not affine in it; without it
        such a face gets placeholder zeros - it is refreshed from the field before it is used)."""
        if self._callable is not None:
            return self._evaluate_callable(t)
        if self.reads_value and value is None:
            return np.zeros(self.face_shape), np.zeros(self.face_shape)
        with np.errstate(all="ignore"):
            v = 0.0 if value is None else np.asarray(value, dtype=np.float64).reshape(self.face_shape)
            a = np.asarray(self._offset(v, self.dx, *self.coords, t), dtype=np.float64)
            b = np.asarray(self._slope(self.dx, *self.coords, t), dtype=np.float64)
        return (np.array(np.broadcast_to(a, self.face_shape), dtype=np.float64, order="C"), np.array(np.broadcast_to(b, self.face_shape), dtype=np.float64, order="C"))


def _c_code(expr) -> str:
    """C source of a sympy expression for the device programs.  Constants such as ``pi`` are printed as their 17-digit values (hiprtc
    sources have no <math.h> macros: ``M_PI`` would not compile); a function the C printer does not know raises
    ``NotImplementedError`` (its fallback is a comment line inside the statement)."""
    from .expr import c_printer

    code = c_printer().doprint(expr)
    if "Not supported in C" in code:
        msg = f"hip backend: `{expr}` has no C form"
        raise NotImplementedError(msg)
    return code


def build_program(lib, entries, info=None) -> Any:
    """One device program (``pdehip_bcprog_create``) that rewrites the coefficient arrays of all ``entries`` —
    ``(face: _AffineFace, const buffer, factor buffer)`` of faces given as sympy expressions — for a time ``t`` (and the field
    the conditions are applied to, layout ``info``: :class:`~pde_hip.device.GridInfo`, for faces that read it):
    returns the handle (``ctypes.c_void_p``), or None when an entry is a Python function (host only)."""
    import ctypes as C

    import sympy as sp

    if not entries or any(face.host_only for face, _, _ in entries):
        return None
    reads = any(face.reads_value for face, _, _ in entries)
    if reads and info is None:
        msg = "boundary conditions that read the field need the layout of the field"
        raise ValueError(msg)
    cases, descs = [], (_abi.BcProgFace * len(entries))()
    for i, (face, buf_a, buf_b) in enumerate(entries):
        offset, slope, syms, value = face._symbolic
        grid = face.grid
        axes = list(grid.axes)
        # the generated function sees the coordinates as c0, c1, c2 in grid-axis order
        sub = {syms["dx"]: sp.Symbol("dx"), syms["t"]: sp.Symbol("t"), value: sp.Symbol("value")}
        for k, name in enumerate(axes):
            sub[syms[name]] = sp.Symbol(f"c{k}")
        cases.append(f"    case {i}: *A = {_c_code(sp.sympify(offset).subs(sub))}; *B = {_c_code(sp.sympify(slope).subs(sub))}; break;")
        d = descs[i]
        d.const_arr, d.factor_arr = buf_a.ptr, buf_b.ptr
        others = [a for a in range(len(axes)) if a != face.axis]
        d.m1 = face.local_shape[others[0]] if len(others) >= 1 else 1
        d.m2 = face.local_shape[others[1]] if len(others) >= 2 else 1
        d.dx = float(grid.discretization[face.axis])
        d.reads_value, d.axis, d.component, d.value_index = int(face.reads_value), int(face.axis), 0, int(face.index)
        bounds = grid.axes_bounds
        for k in range(3):
            d.origin[k], d.step[k], d.index[k], d.first[k] = 0.0, 0.0, 0, 0
        d.origin[face.axis] = float(bounds[face.axis][1] if face.upper else bounds[face.axis][0])   # the wall
        for slot, a in enumerate(others):
            # cell centres (i + 0.5) * dx + x_min, the reference's `discretize_interval` (pde/grids/base.py:88-113)
            d.origin[a], d.step[a], d.index[a], d.first[a] = float(bounds[a][0]), float(grid.discretization[a]), slot + 1, face.lo[a]
    source = ("PDEHIP_BC_FN void bc_face(int face, double value, double dx, double c0, double c1, double c2, double t, double *A, double *B)\n{\n"
              "    (void)value; (void)dx; (void)c0; (void)c1; (void)c2; (void)t;\n    switch (face) {\n" + "\n".join(cases) + "\n    default: break;\n    }\n}\n")
    handle = C.c_void_p()
    lib.bcprog_create(source.encode(), len(entries), descs, getattr(info, "ref", info) if reads else None, C.byref(handle))
    return handle


class BcProgram:
    """Owner of a ``pdehip_bcprog`` handle (destroyed with the object); ``None``-like when the faces cannot run on the device."""

    def __init__(self, lib, entries, info=None):
        self.lib, self.entries = lib, list(entries)
        self.reads_value = any(face.reads_value for face, _, _ in self.entries)
        self.handle = build_program(lib, self.entries, info)

    @property
    def ptr(self):
        return None if self.handle is None else self.handle.value

    def run(self, t: float, stream=None, state=None) -> None:
        """``state``: the (device) field the conditions are applied to - read by faces that are not affine in ``value``."""
        if self.reads_value and state is None:
            msg = "boundary conditions that depend non-linearly on the field need the field they are applied to"
            raise RuntimeError(msg)
        self.lib.bcprog_run(self.handle, float(t), None if state is None else state.ptr, stream)

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None:
            try:
                self.lib.bcprog_destroy(h)
            except Exception:  # noqa: BLE001 - interpreter shutdown
                pass


def program_for(lib, tables, info=None) -> "BcProgram | None":
    """The device program for all faces of ``tables`` (face tables of ONE right-hand side) that must be refreshed - they depend
    on time or read the field (layout ``info``) -, or None when there are none.  Raises ``NotImplementedError`` when a face
    is a Python function (host-probed: no device program)."""
    entries = []
    for tb in {id(tb): tb for tb in tables if tb is not None}.values():
        entries += list(getattr(tb, "_dynamic", []))
    if not entries:
        return None
    prog = BcProgram(lib, entries, info)
    if prog.handle is None:
        msg = "boundary conditions given as Python functions are evaluated on the host"
        raise NotImplementedError(msg)
    return prog


class ExprFaceTable:
    """Face table (``.c`` = ``pdehip_bc_face_t[6]``) whose expression faces can be refreshed for a new time."""

    def __init__(self, table, dynamic: list[tuple[_AffineFace, Any, Any]], write):
        self._table = table
        self.c = table.c
        self.keepalive = table.keepalive
        self._dynamic = dynamic          # (face evaluator, const buffer, factor buffer) of time-dependent faces
        self._write = write
        self._t: float | None = None
        self._program: Any = None        # BcProgram of this table's dynamic faces (built on first use) / False: host evaluation

    @property
    def time_dependent(self) -> bool:
        """Some face has to be refreshed before the conditions are applied (it depends on time or reads the field)."""
        return bool(self._dynamic)

    @property
    def reads_value(self) -> bool:
        """Some face is not affine in the adjacent value: refreshed from the field the conditions are applied to."""
        return any(face.reads_value for face, _, _ in self._dynamic)

    def copy_into(self, dst) -> None:
        self._table.copy_into(dst)

    def _host_values(self, face, state) -> np.ndarray:
        """The value cells of ``face`` out of a full HOST array (test harness: numpy arrays with one ghost layer per axis)."""
        arr = np.asarray(getattr(state, "arr", state))
        return np.take(arr, face.index + 1, axis=face.axis)[(slice(1, -1),) * (arr.ndim - 1)]

    def update(self, args=None, state=None, stream=None) -> None:
        """Re-evaluate the coefficient arrays of the faces that depend on time / on the field ``state`` - the full array the
        conditions are about to be applied to - for ``args["t"]`` (no-op when there are none).  ``stream``: the stream of the
        kernels that consume the arrays (the device program is enqueued there; host-evaluated arrays are copied on it)."""
        if not self._dynamic:
            return
        if args is None or "t" not in args:
            # same contract as the reference (pde/grids/boundaries/local.py:1137-1146): expressions that contain `t` need it,
            # Python functions are evaluated at t = 0
            if any(face.needs_time for face, _, _ in self._dynamic):
                msg = ("Require value for `t` for time-dependent BC. The value must be passed explicitly via `args` when "
                       "calling a differential operator.")
                raise RuntimeError(msg)
            t = 0.0
        else:
            t = float(args["t"])
        reads = self.reads_value
        if reads and state is None:
            msg = "boundary conditions that depend non-linearly on the field need the field they are applied to"
            raise RuntimeError(msg)
        if self._t is not None and t == self._t and self._program is False and not reads:
            return
        if self._program is None:
            # faces given as expressions are refreshed on the device (one launch); Python functions and the host-side tables of
            # the test harness (numpy buffers) are evaluated here
            device = all(not face.host_only and getattr(buf, "arr", None) is None for face, buf, _ in self._dynamic)
            self._program = False
            if device:
                from ._lib import require_device

                prog = BcProgram(require_device(), self._dynamic, getattr(state, "info", None))
                if prog.handle is not None:
                    self._program = prog
        if self._program is not False:
            self._program.run(t, stream=stream, state=state if reads else None)
            self._t = t
            return
        for face, buf_a, buf_b in self._dynamic:
            a, b = face.evaluate(t, self._host_values(face, state) if face.reads_value else None)
            self._write(buf_a, a, stream)
            self._write(buf_b, b, stream)
        self._t = t

    @property
    def host_only(self) -> bool:
        """Some time-dependent face is a Python function: coefficient arrays come from the host before every evaluation."""
        return any(face.host_only for face, _, _ in self._dynamic)


def _write_buffer(buf, arr: np.ndarray, stream=None) -> None:
    host = getattr(buf, "arr", None)
    if host is not None:                      # host-side tables of the test harness
        host[...] = arr.reshape(host.shape)
    else:
        from ._lib import require_device

        require_device().memcpy_h2d(buf.ptr, arr.ctypes.data, arr.nbytes, stream)   # (synchronous for pageable memory: `arr` may go)


def expression_faces(bcs, skip=None) -> dict[tuple[int, bool], Any]:
    """``(axis, upper) -> condition`` of the faces of ``bcs`` that are given as expressions / functions."""
    found: dict[tuple[int, bool], Any] = {}
    if hasattr(bcs, "__iter__"):
        for ax, bc_axis in enumerate(bcs):
            for upper, bc in ((False, bc_axis.low), (True, bc_axis.high)):
                if is_expression_bc(bc) and not (skip and (ax, upper) in skip):
                    found[(ax, upper)] = bc
    return found


def lower_expression_face(bc, table, upload, window=None, part=None):
    """Write the expression face ``bc`` into ``table`` (a first-order face with coefficient arrays, evaluated for t = 0) - cut
    to the box ``window`` of a decomposed grid -; returns its ``(evaluator, const buffer, factor buffer)`` entry when it has
    to be refreshed later (it depends on time or reads the field), else None."""
    face = _AffineFace(bc, window, part)
    a, b = face.evaluate(0.0)
    buf_a, buf_b = upload(a), upload(b)
    table.keepalive += [buf_a, buf_b]
    entry = table.c[2 * face.axis + int(face.upper)]
    entry.kind = _abi.BC_ORDER1
    entry.flags = _abi.BCF_ARRAYS
    entry.index1, entry.index2 = face.index, 0
    entry.const_arr, entry.factor1_arr = buf_a.ptr, buf_b.ptr
    return (face, buf_a, buf_b) if face.time_dependent else None


def convert_bcs_with_expressions(bcs, comp_shape: tuple[int, ...] = (), *, skip=None, upload=None, part=None) -> ExprFaceTable:
    """``convert_bcs`` that also lowers expression conditions onto coefficient arrays (``part``: the table of the real / imaginary part
    of a complex field, see :class:`_AffineFace`)."""
    from .backend import _upload_f64, convert_bcs

    if upload is None:
        upload = _upload_f64
    expr_faces = expression_faces(bcs, skip)
    if part in ("cpl-", "cpl+", "zero") and any(getattr(bc, "_is_func", False) for bc in expr_faces.values()):
        msg = "hip backend: conditions given as Python functions next to conditions with complex factors (on the same operator) are not supported"
        raise NotImplementedError(msg)
    table = convert_bcs(bcs, comp_shape, skip=set(skip or ()) | set(expr_faces), upload=upload, part=part)
    dynamic = []
    for bc in expr_faces.values():
        if comp_shape:
            msg = "Expression boundary conditions only work for scalar conditions"
            raise NotImplementedError(msg)
        entry = lower_expression_face(bc, table, upload, part=part)
        if entry is not None:
            dynamic.append(entry)
    return ExprFaceTable(table, dynamic, _write_buffer)
