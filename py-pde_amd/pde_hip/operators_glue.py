"""Operator glue of the backend: ghost-cell / valid-data / full-data setters, integrators, ``make_operator`` and friends, products of
tensor fields, expressions as functions - :class:`OperatorGlueMixin`.  Split from ``backend.py`` in round 6 (no behaviour change).

Reference: ``pde/backends/numba/backend.py:406-553``, ``pde/backends/numpy/backend.py:72-255``, ``pde/backends/base.py:378-565``.
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
from .evaluation import _ExpressionEvaluation
from .faces import convert_bcs, make_face_setter, real_dtype_of

_logger = logging.getLogger("pde_hip.backend")

# operators that are not linear in their argument: real and imaginary part of a complex field cannot go through them separately
_NONLINEAR_OPERATORS = frozenset({"gradient_squared"})


class OperatorGlueMixin:
    """The operator-facing methods of :class:`~pde_hip.backend.HipBackendMixin`."""

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
        # conditions with complex factors couple the parts (pde_hip/faces.py: convert_bcs): the operator is linear in its ghost cells, the
        # coupling terms are differences of two more applications to the OTHER part
        from .faces import has_complex_factors

        coupling = None
        if has_complex_factors(bcs):
            coupling = {part: convert_bcs_with_expressions(bcs, part=part) if expression_faces(bcs) else convert_bcs(bcs, part=part) for part in ("cpl-", "cpl+", "zero")}
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

            def applied(native, table):
                lib.set_ghost_cells(ginfo.ref, 1, table.c, native.ptr, self.stream)
                res = DeviceArray(ginfo, shape_out[: len(shape_out) - nd])
                op_no_bc(native, res)
                return res.get_valid(stream=self.stream)

            natives = {}
            for part, take in (("re", np.real), ("im", np.imag)):
                native = natives[part] = DeviceArray(ginfo).set_valid(np.ascontiguousarray(take(arr), dtype=real), self.stream)
                if getattr(tables[part], "time_dependent", False):
                    tables[part].update(args, state=native, stream=self.stream)
                parts.append(applied(native, tables[part]))
            if coupling is not None:
                for part, operand in (("cpl-", "im"), ("cpl+", "re"), ("zero", "re")):
                    if getattr(coupling[part], "time_dependent", False):
                        coupling[part].update(args, state=natives[operand], stream=self.stream)
                parts[0] = parts[0] + (applied(natives["im"], coupling["cpl-"]) - applied(natives["im"], coupling["zero"]))
                parts[1] = parts[1] + (applied(natives["re"], coupling["cpl+"]) - applied(natives["re"], coupling["zero"]))
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
