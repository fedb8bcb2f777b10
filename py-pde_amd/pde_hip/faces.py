"""Boundary conditions -> the C face table (``pdehip_bc_face_t[6]``): conversion of a ``BoundariesList`` (mirror or real py-pde), ghost-cell setter
functions on the host, the face setter of `make_full_data_setter`.  Split from ``backend.py`` in round 6 (no behaviour change).

Reference: ``pde/grids/boundaries/local.py:766-1150`` (``get_virtual_point_data``), ``pde/backends/numba/_boundaries.py:256-394``.
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

_logger = logging.getLogger("pde_hip.backend")


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


def has_complex_factors(bcs) -> bool:
    """A condition of ``bcs`` multiplies the field value by a number with an imaginary part (a mixed condition with a complex coefficient):
    real and imaginary part of the ghost cells depend on BOTH parts of the field (see :func:`convert_bcs`)."""
    if not hasattr(bcs, "__iter__"):
        return False
    for bc_axis in bcs:
        for bc in (bc_axis.low, bc_axis.high):
            get = getattr(bc, "get_virtual_point_data", None)
            if type(bc).__name__ in {"ExpressionBC", "ExpressionValueBC", "ExpressionDerivativeBC", "ExpressionMixedBC"}:
                # a condition given as an expression: the slope of its virtual point with respect to `value`
                if getattr(bc, "_is_func", False):
                    continue
                import sympy as sp

                try:
                    expr = sp.sympify(bc.virtual_point_sympy if hasattr(bc, "virtual_point_sympy") else bc._func_expression._sympy_expr)
                except (AttributeError, sp.SympifyError):
                    continue
                value = {sym.name: sym for sym in expr.free_symbols}.get("value")
                if value is None:
                    continue
                slope = sp.diff(expr, value)
                if slope.has(value):
                    continue   # (not affine in `value`: refused where the table is built)
                real = {sym: sp.Symbol(sym.name, real=True) for sym in slope.free_symbols}
                if sp.simplify(sp.expand(slope.xreplace(real)).as_real_imag()[1]) != 0:
                    return True
                continue
            if get is None or type(bc).__name__ in {"UserBC", "_MPIBC"}:
                continue
            data = get()
            factors = (data[1],) if len(data) == 3 else (data[1], data[3])
            if any(np.iscomplexobj(f) and np.any(np.imag(f) != 0) for f in factors):
                return True
    return False


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
    splits into ``Re const + Re factor * Re value`` and ``Im const + Re factor * Im value`` (value, derivative and curvature conditions
    with complex values).  A COMPLEX factor (mixed / Robin conditions with a complex coefficient, ``pde/grids/boundaries/local.py:1927-1938``)
    couples the parts: ``Re ghost = ... - Im factor * Im value``, ``Im ghost = ... + Im factor * Re value``.  Every stencil is linear in its
    ghost cells, so the coupling is the DIFFERENCE of two applications of the operator to the OTHER part (round 6, :func:`has_complex_factors`,
    pde_hip/complex_expr.py): ``part`` "cpl-" / "cpl+" - constant 0, factor ``-/+ Im factor`` on the local faces - and "zero" - constant 0,
    factor 0 on the local faces; periodic axes stay periodic in all of them.
    """
    if upload is None:
        upload = _upload_f64
    if not hasattr(bcs, "__iter__"):
        # BoundariesSetter & co: opaque python callables (pde/grids/boundaries/axes.py:504)
        msg = "hip backend needs a BoundariesList of constant conditions"
        raise NotImplementedError(msg)
    grid = bcs.grid
    table = FaceTable()
    if part not in (None, "re", "im", "cpl-", "cpl+", "zero"):
        msg = f"unknown part `{part}`"
        raise ValueError(msg)
    for ax, bc_axis in enumerate(bcs):
        axis_periodic = bool(getattr(bc_axis, "periodic", False))
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
                if part in ("re", "im"):
                    # (complex factors: the caller adds the coupling terms - has_complex_factors)
                    const = np.real(const) if part == "re" else np.imag(const)
                    f1, f2 = np.real(f1), np.real(f2)
                elif axis_periodic:
                    const, f1, f2 = np.real(const), np.real(f1), np.real(f2)   # (0, +-1: the link to the other side of the axis)
                else:
                    sign = {"cpl-": -1.0, "cpl+": 1.0, "zero": 0.0}[part]
                    const = np.zeros_like(np.real(const))
                    f1, f2 = sign * np.imag(f1) + 0.0, sign * np.imag(f2) + 0.0
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
