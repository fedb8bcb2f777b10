"""Expressions evaluated as FUNCTIONS on device arrays (``make_expression_function``): :class:`_ExpressionEvaluation`.  Split from ``backend.py`` in
round 6 (no behaviour change).  Reference: ``pde/tools/expressions.py`` (``ScalarExpression.get_function``, ``evaluate``).
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

_logger = logging.getLogger("pde_hip.backend")


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
