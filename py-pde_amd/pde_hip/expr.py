"""Expression right-hand sides: sympy expression -> passes of run-time specialised stencil kernels.

Covers the generic ``PDE({"c": "<expression>"})`` class of the reference for a single scalar field
(``pde/pdes/pde.py:299-499``: the reference substitutes operator calls into the expression and lets
numba compile it; temporaries are materialised for every arithmetic operation).  Here the expression
is split into the minimal number of *passes*; each pass is ONE launch of the register-pipelined stencil
kernel compiled (hiprtc, ``csrc/pdehip_jit.hip``) around a generated pointwise epilogue:

    out = f(a, laplace(a), gradient_squared(a), b1, b2, b3; dt, t)        a = the pass' stencil array

so e.g. ``c - c**3 + laplace(c)`` (Allen–Cahn) or ``nu*laplace(h) + lam*gradient_squared(h)`` (KPZ) are a
single pass with 1 read + 1 write per cell, ``laplace(c**3 - c - laplace(c))`` is two passes, and the
Euler update / RK stage scaling is folded into the last pass.  Supported: ``laplace``,
``gradient_squared`` (central) and the per-axis derivatives ``d_dx`` / ``d2_dx2`` ... (central; Burgers
``-u*d_dx(u) + nu*d2_dx2(u)``, KdV ``-6*u*d_dx(u) - d_dx(d2_dx2(u))``) of the field or of any pointwise
sub-expression, elementary functions, constants, explicit time ``t``.  Anything else raises ``NotImplementedError`` like the reference does
for unknown backends (``pde/pdes/pde.py:469-496``).
"""

from __future__ import annotations

import ctypes as C
from typing import Any

import numpy as np

from . import _abi

MAX_EXTRA = 3
OPERATORS = ("laplace", "gradient_squared")
P_DT, P_T = 0, 1  # slots of the run-time parameter vector
P_FIRST_REDUCTION, MAX_REDUCTIONS = 2, 8   # p[2..9]: integrals over the grid (reduction passes)



def c_printer():
    """The C printer of the generated device code (right-hand-side epilogues here, boundary programs in bc_expr.py): small integer powers
    as repeated multiplication (numba does the same), constants such as ``pi`` / ``E`` as 17-digit literals - the run-time compiled
    sources include no <math.h>, so ``M_PI`` would not compile; numpy, which the reference evaluates with, uses the same
    double values."""
    from sympy.printing.c import C99CodePrinter

    class Printer(C99CodePrinter):
        def _print_Pow(self, e):
            b, ex = e.as_base_exp()
            if ex.is_Integer and 1 <= abs(int(ex)) <= 8:
                prod = "*".join([f"({self._print(b)})"] * abs(int(ex)))
                return f"({prod})" if ex > 0 else f"(1.0/({prod}))"
            return super()._print_Pow(e)

        def _print_NumberSymbol(self, e):
            return repr(float(e.evalf(17)))

        _print_Pi = _print_Exp1 = _print_EulerGamma = _print_GoldenRatio = _print_Catalan = _print_TribonacciConstant = _print_NumberSymbol

    return Printer({"precision": 17})

class _Pass:
    """One kernel launch: stencil array `src`, centre-only arrays `extras`, result array `out`."""

    def __init__(self, src: str, extras: list[str], out: str, expr, reduce_slot: int | None = None):
        self.src, self.extras, self.out, self.expr = src, extras, out, expr
        self.reduce_slot = reduce_slot   # not None: no kernel of its own - `integral(src)` goes into parameter p[slot]


def _sympy():
    import sympy

    return sympy


class ExpressionPlan:
    """Lower ``expr`` (string, variable ``var``) into passes.  Array names: ``"state"`` (the equation's own variable),
    ``"var:<name>"`` (the other scalar fields of a multi-field PDE, ``others``), ``"tmp<k>"``."""

    def __init__(self, expr_str: str, var: str, consts: dict[str, Any] | None = None, others: tuple[str, ...] = (),
                 axes: tuple[str, ...] = (), aliases: dict[str, str] | None = None, aux: tuple[str, ...] = (),
                 vectors: dict[str, tuple[str, ...]] | None = None, component: int | tuple[int, int] | None = None,
                 user_funcs: dict[str, Any] | None = None, tensors: dict[str, tuple[tuple[str, ...], ...]] | None = None):
        """``axes``: the grid's axis names (``grid.axes``); they name the per-axis derivatives ``d_d<ax>`` / ``d2_d<ax>2``
        (central; reference: numba/backend.py:105-173) that an expression may use besides OPERATORS.  ``aliases``: further
        operator names standing for one of OPERATORS (``{"laplace_outer": "laplace"}``) - same stencil, but a name of its own
        and therefore boundary conditions of its own (PDE classes whose nested operators take different conditions, e.g.
        ``bc`` / ``bc_lap`` of pde/pdes/swift_hohenberg.py:104-105).  ``aux``: names that stand for further arrays on the
        grid which the caller supplies (array-valued ``consts``, the cell coordinates ``x``, ``y``, ``z`` of expressions that
        depend on position: pde/pdes/pde.py:441-447); they enter a pass as centre-only inputs (array name ``aux:<name>``).
        ``vectors``: vector FIELDS of the state by name -> the names of their scalar components among ``var`` / ``others``
        (``{"u": ("u#0", "u#1")}``); ``tensors``: rank-2 FIELDS of the state -> rows of component names (``{"S": (("S#0#0", "S#0#1"),
        ("S#1#0", "S#1#1"))}``); ``component``: the plan evaluates this component of a vector-valued (``k``) or tensor-valued
        (``(i, j)``) right-hand side (the equation of a vector / tensor field; ``var`` is then the name of that component of the field).  ``user_funcs``: the Python
        functions of ``pde.PDE(..., user_funcs=...)`` (pde/pdes/pde.py:84, pde/tools/expressions.py:173-212).  They cannot run on
        the device as Python; they are TRACED once with symbolic arguments (scalars as sympy expressions, vectors / tensors as
        numpy object arrays of them) and what they return is compiled like the rest of the expression - which covers
        arithmetic, indexing, numpy reductions over components and sympy functions; anything else raises
        ``NotImplementedError``."""
        sp = _sympy()
        self.var = var
        self.user_funcs = dict(user_funcs or {})
        self.others = tuple(others)
        expr_str = expr_str.replace("∇²", "laplace").replace("^", "**")
        # operator name -> ("d1" | "d2", normalised axis) for the per-axis derivatives; the kernels number axes right-aligned to 3
        self.axis_ops: dict[str, tuple[str, int]] = {}
        for k, ax in enumerate(axes):
            self.axis_ops[f"d_d{ax}"] = ("d1", 3 - len(axes) + k)
            self.axis_ops[f"d2_d{ax}2"] = ("d2", 3 - len(axes) + k)
        self.aliases = dict(aliases or {})
        # components of the vector operators `gradient` / `divergence` (central), lowered to per-axis atoms by `_lower_vectors`
        nd = len(axes)
        self.vector_ops = {}
        for k in range(nd):
            self.vector_ops[f"grad_{k}"] = ("gr", 3 - nd + k)   # k-th component of gradient(s): conditions of `gradient`
            self.vector_ops[f"div_{k}"] = ("gr", 3 - nd + k)    # k-th term of divergence(v): component k of the vector conditions
            self.aliases[f"vlap_{k}"] = "laplace"               # k-th component of vector_laplace(v): ... of `vector_laplace`
            self.vector_ops[f"vlap_{k}"] = ("lap", -1)
            for j in range(nd):
                # vector_gradient(v)[k][j] = d_j v_k with component k of the conditions of `vector_gradient`;
                # tensor_divergence(T)[k] = sum_j d_j T[k][j] with component (k, j) of the rank-2 conditions (cartesian.py:999-1096)
                self.vector_ops[f"vgrad_{k}_{j}"] = ("gr", 3 - nd + j)
                self.vector_ops[f"tdiv_{k}_{j}"] = ("gr", 3 - nd + j)
        for alias, base in list(self.aliases.items()):
            if base in self.vector_ops and alias not in self.vector_ops:
                # another name of a per-axis atom of a vector operator (`grad_0_imop`: the imaginary operand, complex_expr.py)
                self.vector_ops[alias] = self.vector_ops[base]
                if self.vector_ops[base][0] == "lap":
                    self.aliases[alias] = "laplace"
                else:
                    del self.aliases[alias]
            elif base in self.axis_ops:    # another name of a per-axis derivative (its own conditions: complex_expr.py)
                self.axis_ops[alias] = self.axis_ops[base]
                del self.aliases[alias]
            elif base not in OPERATORS:
                msg = f"operator alias `{alias}` must stand for one of {OPERATORS} or a per-axis derivative"
                raise ValueError(msg)
        self.axis_ops.update({k: v for k, v in self.vector_ops.items() if v[0] != "lap"})
        self._ops = {name: sp.Function(name) for name in (*OPERATORS, *self.axis_ops, *self.aliases)}
        # `integral(f)`: the integral of a (pointwise) expression over the grid, a number (pde/pdes/pde.py:355-373 takes the
        # operator of that name from the grid); evaluated by a reduction pass, it reaches the kernels as a run-time parameter
        self._integral = sp.Function("integral")
        self.reductions: list[str] = []   # array integrated into parameter slot P_FIRST_REDUCTION + index
        local: dict[str, Any] = dict(self._ops)
        local["integral"] = self._integral
        self._state = sp.Symbol("__state", real=True)  # internal name: must not clash with the code symbols
        self._t = sp.Symbol("__t", real=True)
        local[var] = self._state
        local["t"] = self._t
        other_syms = {name: sp.Symbol(f"__v_{name}", real=True) for name in self.others}
        local.update(other_syms)
        aux_syms = {name: sp.Symbol(f"__a_{name}", real=True) for name in aux if name != var and name not in self.others}
        local.update(aux_syms)
        # vector fields of the state: a marker symbol per field, replaced by its components in `_lower_vectors`
        self._vector_fields: dict[Any, list] = {}
        for vname, comps in (vectors or {}).items():
            marker = sp.Symbol(f"__vec_{vname}", real=True)
            local[vname] = marker
            self._vector_fields[marker] = [self._state if c == var else other_syms[c] for c in comps]
        self._tensor_fields: dict[Any, list] = {}
        for tname, rows in (tensors or {}).items():
            marker = sp.Symbol(f"__ten_{tname}", real=True)
            local[tname] = marker
            self._tensor_fields[marker] = [[self._state if c == var else other_syms[c] for c in row] for row in rows]
        for k, v in (consts or {}).items():
            if k in aux_syms:
                continue
            if not np.isscalar(v):
                msg = f"hip backend: constant `{k}` is an array but was not announced as one (internal)"
                raise NotImplementedError(msg)
            local[k] = sp.Float(float(v))
        if nd:
            local.update({name: sp.Function(name) for name in ("gradient", "divergence", "dot", "inner", "vector_laplace", "vector_gradient",
                                                                "tensor_divergence", "outer")})
        for name in self.user_funcs:
            local[name] = sp.Function(name)
        try:
            expr = sp.sympify(expr_str, locals=local)
        except (sp.SympifyError, SyntaxError, TypeError) as err:
            msg = f"cannot parse expression `{expr_str}`: {err}"
            raise ValueError(msg) from err
        if nd or self.user_funcs:
            kind, expr = self._lower_vectors(expr, nd)
            if component is None and kind != "s":
                msg = f"hip backend: the right-hand side `{expr_str}` is a vector, the field is a scalar"
                raise NotImplementedError(msg)
            if isinstance(component, tuple):
                if kind != "t":
                    msg = f"the right-hand side `{expr_str}` of a tensor field must be a tensor"
                    raise ValueError(msg)
                expr = expr[component[0]][component[1]]
            elif component is not None:
                if kind != "v":
                    msg = f"the right-hand side `{expr_str}` of a vector field must be a vector"
                    raise ValueError(msg)
                expr = expr[component]
        unknown = {f.func.__name__ for f in expr.atoms(sp.core.function.AppliedUndef)} - set(self._ops) - {"integral"}
        if unknown:
            msg = f"hip backend has no kernel for operator(s) {sorted(unknown)} in `{expr_str}`"
            raise NotImplementedError(msg)
        free = {str(s) for s in expr.free_symbols} - {"__state", "__t"} - {str(v) for v in other_syms.values()} - {str(v) for v in aux_syms.values()}
        if free:
            msg = f"unknown symbol(s) {sorted(free)} in `{expr_str}` (pass them in `consts`)"
            raise ValueError(msg)
        self.uses_time = self._t in expr.free_symbols
        self.operators_used = sorted({f.func.__name__ for f in expr.atoms(sp.core.function.AppliedUndef)} - {"integral"})
        self.passes: list[_Pass] = []
        self._ntmp = 0
        self._memo: dict[Any, str] = {}
        self._arrays = {"state": self._state}  # array name -> sympy symbol standing for its centre value
        for name, sym in other_syms.items():
            self._arrays[f"var:{name}"] = sym
        self.aux_used = tuple(name for name, sym in aux_syms.items() if sym in expr.free_symbols)
        for name in self.aux_used:
            self._arrays[f"aux:{name}"] = aux_syms[name]
        self._lower_top(expr)

    def _lower_vectors(self, e, nd: int):
        """Vector-valued sub-expressions component by component: ``gradient(s)`` -> ``[grad_k(s)]``, scalar * vector, sums of
        vectors, ``dot(v, w)`` -> ``sum v_k w_k``, ``divergence(v)`` -> ``sum div_k(v_k)`` (central differences like
        pde/backends/numba/operators/cartesian.py:386-587, :812-996).  Returns ``("s", expr)`` or ``("v", [components])``."""
        sp = _sympy()
        if isinstance(e, sp.core.function.AppliedUndef):
            name = e.func.__name__
            args = [self._lower_vectors(a, nd) for a in e.args]
            if name in self.user_funcs:
                return self._trace_user_func(name, args, nd)
            if name == "gradient":
                if len(args) != 1 or args[0][0] != "s":
                    msg = "hip backend: `gradient` inside expressions takes one scalar argument"
                    raise NotImplementedError(msg)
                return "v", [self._ops[f"grad_{k}"](args[0][1]) for k in range(nd)]
            if name == "divergence":
                if len(args) != 1 or args[0][0] != "v":
                    msg = "`divergence` needs a vector argument"
                    raise ValueError(msg)
                return "s", sp.Add(*[self._ops[f"div_{k}"](c) for k, c in enumerate(args[0][1])])
            if name == "vector_laplace":
                if len(args) != 1 or args[0][0] != "v":
                    msg = "`vector_laplace` needs a vector argument"
                    raise ValueError(msg)
                return "v", [self._ops[f"vlap_{k}"](c) for k, c in enumerate(args[0][1])]
            if name == "vector_gradient":
                if len(args) != 1 or args[0][0] != "v":
                    msg = "`vector_gradient` needs a vector argument"
                    raise ValueError(msg)
                return "t", [[self._ops[f"vgrad_{i}_{j}"](c) for j in range(nd)] for i, c in enumerate(args[0][1])]
            if name == "tensor_divergence":
                if len(args) != 1 or args[0][0] != "t":
                    msg = "`tensor_divergence` needs a tensor argument"
                    raise ValueError(msg)
                return "v", [sp.Add(*[self._ops[f"tdiv_{i}_{j}"](c) for j, c in enumerate(row)]) for i, row in enumerate(args[0][1])]
            if name == "outer":
                if len(args) != 2 or args[0][0] != "v" or args[1][0] != "v":
                    msg = "`outer` needs two vector arguments"
                    raise ValueError(msg)
                return "t", [[a * b for b in args[1][1]] for a in args[0][1]]
            if name in ("dot", "inner"):
                # pde/backends/numpy/backend.py:306-335: vector . vector, tensor . vector, vector . tensor, tensor . tensor
                if len(args) != 2 or args[0][0] == "s" or args[1][0] == "s":
                    msg = "Fields in dot product must have rank >= 1"
                    raise TypeError(msg)
                (ka, a), (kb, b) = args
                if ka == "v" and kb == "v":
                    return "s", sp.Add(*[x * y for x, y in zip(a, b)])
                if ka == "t" and kb == "v":
                    return "v", [sp.Add(*[a[i][j] * b[j] for j in range(nd)]) for i in range(nd)]
                if ka == "v" and kb == "t":
                    return "v", [sp.Add(*[a[i] * b[i][j] for i in range(nd)]) for j in range(nd)]
                return "t", [[sp.Add(*[a[i][j] * b[j][k] for j in range(nd)]) for k in range(nd)] for i in range(nd)]
            if any(k != "s" for k, _ in args):
                msg = f"hip backend: operator `{name}` of a vector / tensor inside expressions is not supported"
                raise NotImplementedError(msg)
            return "s", e.func(*[a for _, a in args])
        if not e.args:
            if e in self._vector_fields:
                return "v", list(self._vector_fields[e])
            if e in self._tensor_fields:
                return "t", [list(row) for row in self._tensor_fields[e]]
            return "s", e
        parts = [self._lower_vectors(a, nd) for a in e.args]
        if all(k == "s" for k, _ in parts):
            return "s", e.func(*[a for _, a in parts])
        ranks = {k for k, _ in parts}
        if e.is_Add:
            if len(ranks) != 1:
                msg = "cannot add fields of different rank"
                raise ValueError(msg)
            if ranks == {"v"}:
                return "v", [sp.Add(*[p[1][k] for p in parts]) for k in range(nd)]
            return "t", [[sp.Add(*[p[1][i][j] for p in parts]) for j in range(nd)] for i in range(nd)]
        if e.is_Mul and sum(k != "s" for k, _ in parts) == 1:
            kind, val = next(p for p in parts if p[0] != "s")
            scal = sp.Mul(*[p[1] for p in parts if p[0] == "s"])
            if kind == "v":
                return "v", [scal * c for c in val]
            return "t", [[scal * c for c in row] for row in val]
        msg = f"hip backend: vector expression `{e}` is not supported (sums, scalar multiples, dot, divergence)"
        raise NotImplementedError(msg)

    def _trace_user_func(self, name: str, args, nd: int):
        """Call the user's Python function ONCE with symbolic arguments and lower what it returns."""
        sp = _sympy()

        def to_py(kind, val):
            return val if kind == "s" else np.array(val, dtype=object)

        try:
            result = self.user_funcs[name](*[to_py(k, v) for k, v in args])
        except Exception as err:   # noqa: BLE001 - whatever the user's code raises on symbolic input
            msg = (f"hip backend: user function `{name}` cannot be traced symbolically ({type(err).__name__}: {err}); functions that run "
                   "on the device must work on sympy expressions (arithmetic, indexing, sympy functions)")
            raise NotImplementedError(msg) from err
        arr = np.asarray(result, dtype=object)
        try:
            if arr.ndim == 0:
                return "s", sp.sympify(arr.item())
            if arr.shape == (nd,):
                return "v", [sp.sympify(c) for c in arr]
            if arr.shape == (nd, nd):
                return "t", [[sp.sympify(c) for c in row] for row in arr]
        except (sp.SympifyError, TypeError) as err:
            msg = f"hip backend: user function `{name}` returned something that is not an expression: {err}"
            raise NotImplementedError(msg) from err
        msg = f"hip backend: user function `{name}` returned an array of shape {arr.shape}"
        raise NotImplementedError(msg)

    # --- lowering --------------------------------------------------------------------------------
    def _new_tmp(self) -> str:
        sp = _sympy()
        name = f"tmp{self._ntmp}"
        self._ntmp += 1
        self._arrays[name] = sp.Symbol(f"__{name}", real=True)
        return name

    def _materialise(self, expr) -> str:
        """Name of the temporary holding the (operator-lowered) pointwise expression ``expr``;
        identical sub-expressions share one temporary."""
        if expr not in self._memo:
            tmp = self._new_tmp()
            self._emit(expr, tmp)
            self._memo[expr] = tmp
        return self._memo[expr]

    def _array_of(self, sym) -> str | None:
        for name, s in self._arrays.items():
            if s == sym:
                return name
        return None

    def _lower_ops(self, expr):
        """Replace every operator application by an atom ``op(array_symbol)``; operator arguments that
        are not plain arrays are materialised into temporaries first (innermost first)."""
        sp = _sympy()
        if isinstance(expr, sp.core.function.AppliedUndef):
            (arg,) = expr.args
            arg = self._lower_ops(arg)
            if self._array_of(arg) is None:
                arg = self._arrays[self._materialise(arg)]
            if expr.func == self._integral:
                name = self._array_of(arg)
                if name not in self.reductions:
                    if len(self.reductions) >= MAX_REDUCTIONS:
                        msg = f"hip backend: more than {MAX_REDUCTIONS} integrals in one expression"
                        raise NotImplementedError(msg)
                    self.reductions.append(name)
                    self.passes.append(_Pass(name, [], f"integral{len(self.reductions) - 1}", None, P_FIRST_REDUCTION + len(self.reductions) - 1))
                return sp.Symbol(f"p[{P_FIRST_REDUCTION + self.reductions.index(name)}]", real=True)
            from .complex_expr import COUPLING_SUFFIXES

            if expr.func.__name__.endswith(tuple(COUPLING_SUFFIXES)):
                # the coupling terms of complex-factor conditions (complex_expr.py): the same stencil with OTHER face tables than the operator's
                # own application to this array - a pass takes one table, so each of them is a pass (and a temporary) of its own
                return self._arrays[self._materialise(expr.func(arg))]
            return expr.func(arg)
        if expr.args:
            return expr.func(*[self._lower_ops(a) for a in expr.args])
        return expr

    def _emit(self, expr, out: str) -> None:
        """Emit the passes that evaluate the (operator-lowered) pointwise expression into ``out``."""
        sp = _sympy()
        # (sorted: the order of a set of sympy atoms follows the hash seed of the process - the pass order must not, every rank of a
        # decomposed run builds the same plan and exchanges the operands in the same order)
        atoms = sorted(expr.atoms(sp.core.function.AppliedUndef), key=sp.default_sort_key)
        by_array: dict[str, list] = {}
        for a in atoms:
            by_array.setdefault(self._array_of(a.args[0]), []).append(a)
        # the stencil array of this pass: the one that leaves the fewest operator atoms (on other arrays,
        # not yet materialised) to be computed by passes of their own; ties prefer the latest temporary
        if by_array:
            names = list(by_array)

            def cost(name):
                return sum(1 for other, lst in by_array.items() if other != name for a in lst if a not in self._memo)

            # (ties in the pass that writes the result: the equation's own variable - the Euler update reads it anyway, as stencil array it costs no slot)
            src = min(reversed(names), key=lambda name: (cost(name), 0 if (out == "out" and name == "state") else 1))
        else:
            used = [n for n, s in self._arrays.items() if s in expr.free_symbols]
            src = used[0] if used else "state"
        # operator atoms on OTHER arrays become (or re-use) temporaries
        for name, lst in by_array.items():
            if name == src:
                continue
            for a in lst:
                expr = expr.subs(a, self._arrays[self._materialise(a)])
        def extras_of(e):
            return [n for n, s in self._arrays.items() if n != src and s in e.free_symbols]

        # a pass reads its stencil array and at most MAX_EXTRA others: pointwise terms of a sum that need more arrays are
        # evaluated by passes of their own (term by term, the one with the most arrays first) until the rest fits
        def arrays_in(e):
            return {n for n, s in self._arrays.items() if s in e.free_symbols}

        def limit():
            # the pass that writes the result also carries the Euler update `state + dt * F`: when the state is neither its
            # stencil array nor one of its inputs, one slot stays free for it
            need_state = out == "out" and src != "state" and "state" not in extras_of(expr)
            return MAX_EXTRA - (1 if need_state else 0)

        while len(extras_of(expr)) > limit():
            # Somewhere in the tree a sum or a product has pointwise arguments (no operators left in them) that can be
            # evaluated by a pass of their own: the partial sum / product over a greedily chosen group of them replaces the
            # group's arrays by one temporary.  The node whose group covers the most arrays goes first.
            best = None
            for node in sp.preorder_traversal(expr):
                if not (node.is_Add or node.is_Mul):
                    continue
                terms = [t for t in node.args if not t.atoms(sp.core.function.AppliedUndef) and arrays_in(t)]
                terms.sort(key=lambda t: -len(arrays_in(t)))
                group, used = [], set()
                for t in terms:
                    if len(used | arrays_in(t)) <= MAX_EXTRA:   # (its own pass: stencil array + extras, a slot to spare)
                        group.append(t)
                        used |= arrays_in(t)
                if len(used) < 2 or (len(group) == 1 and self._array_of(group[0]) is not None):
                    continue
                if best is None or len(used) > best[0]:
                    best = (len(used), node, group)
            if best is None:
                msg = "hip backend: expression needs more than 3 auxiliary fields in one pass"
                raise NotImplementedError(msg)
            _, node, group = best
            partial = node.func(*group)
            rest = [t for t in node.args if not any(t is g for g in group)]
            expr = expr.xreplace({node: node.func(*rest, self._arrays[self._materialise(partial)])})
        extras = extras_of(expr)
        self.passes.append(_Pass(src, extras, out, expr))

    def _lower_top(self, expr) -> None:
        self._emit(self._lower_ops(expr), "out")

    # --- code generation ---------------------------------------------------------------------------
    def epilogue(self, p: _Pass, wrap: str) -> tuple[str, list[str]]:
        """C body of ``pde_epilogue`` for pass ``p`` and the arrays bound to e0..e2.

        ``wrap`` in {"rate", "scaled", "euler"} selects what the LAST pass returns: F, dt*F, state + dt*F.
        """
        sp = _sympy()
        c, lap, gsq = sp.symbols("c lap gsq", real=True)
        e_syms = sp.symbols("e0 e1 e2", real=True)
        sub: dict[Any, Any] = {}
        src_sym = self._arrays[p.src]
        for a in p.expr.atoms(sp.core.function.AppliedUndef):
            name = a.func.__name__
            if name in self.axis_ops:
                which, axis = self.axis_ops[name]
                sub[a] = sp.Symbol(f"d.{which}[{axis}]", real=True)   # PdeDer of the kernels (csrc/pdehip_device.h)
            else:
                sub[a] = lap if self.aliases.get(name, name) == "laplace" else gsq
        sub2 = dict(sub)
        sub2[src_sym] = c
        sub2[self._t] = sp.Symbol("t", real=True)
        for i, name in enumerate(p.extras):
            sub2[self._arrays[name]] = e_syms[i]
        expr = p.expr.subs(sub2, simultaneous=True)
        extras = list(p.extras)
        code = c_printer().doprint(expr)
        if "Not supported in C" in code:
            msg = f"hip backend: `{expr}` has no C form"
            raise NotImplementedError(msg)
        lines = [f"const double t = p[{P_T}]; (void)t;", f"const double F = {code};"]
        if p.out != "out" or wrap == "rate":
            lines.append("return F;")
        elif wrap == "scaled":
            lines.append(f"return p[{P_DT}] * F;")
        elif wrap == "euler":
            if p.src == "state":
                y = "c"
            else:
                if "state" not in extras:
                    if len(extras) >= MAX_EXTRA:
                        msg = "hip backend: no slot left for the state in the Euler update"
                        raise NotImplementedError(msg)
                    extras.append("state")
                y = f"e{extras.index('state')}"
            lines.append(f"return {y} + p[{P_DT}] * F;")
        else:
            raise ValueError(wrap)
        return "\n".join(lines), extras

    def describe(self) -> list[str]:
        return [f"p[{p.reduce_slot}] <- integral({p.src})" if p.reduce_slot is not None else
                f"{p.out} <- f({p.src}; ops={sorted({a.func.__name__ for a in p.expr.atoms(_sympy().core.function.AppliedUndef)})}; extras={p.extras})"
                for p in self.passes]


def _run_loop(lib, info, loop, state, other, ncomp: int, dt: float, t0: float, nsteps: int, uses_time: bool, stream, program=None):
    passes, fixed, nfixed, _keep = loop
    result = C.c_void_p()
    lib.jit_euler_run(info.ref, passes, len(passes), fixed, nfixed, state.ptr, other.ptr, ncomp, dt, t0, int(bool(uses_time)), int(nsteps),
                      None if program is None else program.ptr, C.byref(result), stream)
    return state if result.value == state.ptr else other


def _run_rk(lib, info, loop, ncomp: int, y, ynew, work, err, dt: float, t0: float, nsteps: int, ctl, stage_fuse: bool, stream, program=None,
            euler_adaptive: bool = False):
    """``pdehip_jit_rk_run``: ``nsteps`` RK4 steps in place on ``y`` (``ctl`` None) or the adaptive RKF45 loop described by ``ctl``;
    ``euler_adaptive``: the reference's adaptive Euler loop instead (``pdehip_jit_euler_adaptive_run``, work = rate, half step, scratch).
    Returns the array that holds the final state."""
    from .device import ptr_array

    passes, fixed, nfixed, _keep = loop
    result = C.c_void_p()
    if euler_adaptive:
        lib.jit_euler_adaptive_run(info.ref, passes, len(passes), fixed, nfixed, ncomp, y.ptr, ynew.ptr, ptr_array(work), err.ptr, C.byref(ctl),
                                   int(stage_fuse), None if program is None else program.ptr, C.byref(result), stream)
        return y if result.value == y.ptr else ynew
    lib.jit_rk_run(info.ref, passes, len(passes), fixed, nfixed, ncomp, y.ptr, None if ynew is None else ynew.ptr, ptr_array(work),
                   None if err is None else err.ptr, float(dt), float(t0), int(nsteps), None if ctl is None else C.byref(ctl),
                   int(stage_fuse), None if program is None else program.ptr, C.byref(result), stream)
    return y if result.value == y.ptr else ynew


class ExpressionRhs:
    """Device evaluation of an :class:`ExpressionPlan` (kernels compiled lazily, cached per wrap mode)."""

    def __init__(self, backend, plan: ExpressionPlan, info, tables: dict, aux: dict | None = None):
        """``aux``: device arrays of the plan's auxiliary inputs by name (array-valued constants, cell coordinates).
        ``tables``: operator name -> face table.  The reference applies ONE boundary condition per operator name to
        every application of that operator, nested ones included (``pde/pdes/pde.py:329-343``); a pass therefore takes
        the table of the operator(s) it evaluates.  Operators with equal conditions share one table object."""
        from .device import DeviceArray

        self.backend, self.plan, self.info = backend, plan, info
        self.lib = backend._lib
        self.tables = tables
        self.aux = {f"aux:{name}": arr for name, arr in (aux or {}).items() if name in plan.aux_used}
        missing = [name for name in plan.aux_used if f"aux:{name}" not in self.aux]
        if missing:
            msg = f"no array supplied for {missing}"
            raise ValueError(msg)
        self.pass_faces = []
        sp = _sympy()
        for p in plan.passes:
            if p.reduce_slot is not None:
                self.pass_faces.append(None)
                continue
            ops = sorted({a.func.__name__ for a in p.expr.atoms(sp.core.function.AppliedUndef)})
            distinct = {id(tables[o]): tables[o] for o in ops}
            if len(distinct) > 1:
                msg = f"hip backend: operators {ops} with different boundary conditions on the same field cannot share one sweep"
                raise NotImplementedError(msg)
            self.pass_faces.append(next(iter(distinct.values())) if distinct else None)
        self.tmps = {}
        for p in plan.passes:
            if p.out != "out" and p.reduce_slot is None:
                self.tmps[p.out] = DeviceArray(info)
        self.has_reductions = bool(plan.reductions)
        if self.has_reductions:
            from .device import DeviceBuffer

            self._red_dev = DeviceBuffer(8)
            self._red_host = np.zeros(1)
            self._cell_volume = float(np.prod(info.dx))
        self._dynamic = [tb for tb in {id(tb): tb for tb in tables.values()}.values() if getattr(tb, "time_dependent", False)]
        # conditions that are not affine in the adjacent value read it from the field they are applied to: here that has to be
        # the state itself (the values of intermediate fields are not known when the conditions are refreshed)
        # ... an operator of a nested expression applies them to an INTERMEDIATE field (like the reference: `value` is the adjacent
        # value of whatever the operator acts on): those conditions are refreshed pass by pass from the pass's own input, which
        # keeps the evaluation out of the C loops and of the fused two-level sweep
        self._reads_intermediate = any(tb is not None and getattr(tb, "reads_value", False) and p.src != "state"
                                       for p, tb in zip(plan.passes, self.pass_faces))
        if any(getattr(tb, "reads_value", False) for tb in self._dynamic) and getattr(plan, "component", None) is not None:
            msg = "hip backend: boundary conditions that depend non-linearly on the field, for the components of a vector field"
            raise NotImplementedError(msg)
        self._kernels: dict[tuple[int, str], tuple[C.c_void_p, list[str]]] = {}
        # decomposed grids (pde_hip/distributed.py): `_exchange(array)` fills the ghost layers towards the neighbouring ranks before a
        # pass applies operators to `array`; the passes then run one by one from Python (no fused chains, no C loops)
        self._exchange = None
        self._reduce = None
        self._pass_by_pass = False
        # two steps per sweep: the second level would need the faces at t + dt / the integrals of the intermediate level
        self._two_ok: bool | None = False if (self._dynamic or self.has_reductions) else None
        self._fused: dict[str, C.c_void_p | None] = {}

    def _update_faces(self, t: float, state=None) -> None:
        """Coefficient arrays of faces that depend on the time or on the field ``state`` (expression BCs, ``pde_hip/bc_expr.py``)."""
        for tb in self._dynamic:
            tb.update({"t": t}, state=state, stream=self.backend.stream)

    def _faces(self, index: int):
        t = self.pass_faces[index]
        return None if t is None else t.c

    def _refresh_for_pass(self, index: int, src, t: float) -> None:
        """Conditions that read the field, applied to an intermediate field: rewritten from the input of THIS pass."""
        tb = self.pass_faces[index]
        if self._reads_intermediate and tb is not None and getattr(tb, "reads_value", False):
            tb.update({"t": t}, state=src, stream=self.backend.stream)     # (also for passes on the state itself: an earlier pass may have rewritten a shared table)

    def _kernel(self, index: int, wrap: str):
        key = (index, wrap if self.plan.passes[index].out == "out" else "rate")
        if key not in self._kernels:
            p = self.plan.passes[index]
            body, extras = self.plan.epilogue(p, key[1])
            h = C.c_void_p()
            self.lib.jit_create(body.encode(), C.byref(h))
            self._kernels[key] = (h, extras)
        return self._kernels[key]

    def check(self, dtype, ndim: int) -> None:
        """Compile every kernel (no device needed) — surfaces code-generation errors early."""
        for i in range(len(self.plan.passes)):
            if self.plan.passes[i].reduce_slot is not None:
                continue
            for wrap in ("rate", "scaled", "euler"):
                h, _ = self._kernel(i, wrap)
                self.lib.jit_check(h, _abi.dtype_code(dtype), ndim)
        for wrap in ("rate", "scaled", "euler"):   # the fused two-level kernel of a two-pass chain
            h = self._fused_handle(wrap)
            if h is not None:
                self.lib.jit_check(h, _abi.dtype_code(dtype), ndim)

    def apply(self, state, out, wrap: str = "rate", dt: float = 0.0, t: float = 0.0, others: dict | None = None) -> None:
        """out = F(state)  |  dt*F(state)  |  state + dt*F(state)   (wrap = rate | scaled | euler); ``others``: the other
        fields of a multi-field PDE by variable name."""
        arrays = {"state": state, "out": out, **self.tmps, **self.aux}
        for name, arr in (others or {}).items():
            arrays[f"var:{name}"] = arr
        nparams = P_FIRST_REDUCTION + len(self.plan.reductions)
        params = (C.c_double * nparams)(dt, t)
        self._update_faces(t, state)
        if self._fused2(state, out, wrap, params):
            return
        exchanged: set[str] = set()
        for i, p in enumerate(self.plan.passes):
            if p.reduce_slot is not None:
                # integral over the grid -> run-time parameter of the passes that follow (8 bytes cross PCIe: a host sync)
                self.lib.integrate(self.info.ref, 1, arrays[p.src].ptr, self._cell_volume, self._red_dev.ptr, self.backend.stream)
                self.lib.memcpy_d2h(self._red_host.ctypes.data, self._red_dev.ptr, 8, self.backend.stream)
                # (decomposed grids: the integral over the box of this rank; `_reduce` sums over the ranks, like the reference's
                # `mpi_allreduce` in `ScalarField.integral`, pde/fields/datafield_base.py)
                params[p.reduce_slot] = float(self._red_host[0]) if self._reduce is None else float(self._reduce(float(self._red_host[0])))
                continue
            h, extras = self._kernel(i, wrap)
            ex = (C.c_void_p * 3)()
            for m, name in enumerate(extras):
                ex[m] = arrays[name].ptr
            self._refresh_for_pass(i, arrays[p.src], t)
            if self._exchange is not None and self.pass_faces[i] is not None and p.src not in exchanged:
                self._exchange(arrays[p.src])      # (once per evaluation: no array is written after it has been read by a pass)
                exchanged.add(p.src)
            self.lib.jit_apply(h, self.info.ref, arrays[p.src].ptr, ex, arrays[p.out].ptr, params, nparams, self._faces(i), self.backend.stream)

    def apply_stage(self, state, k_out, dt: float, t: float, kind: int, y, ks, coefs, c_new: float, out2, err=None) -> bool:
        """k = dt*F(state) and, in the same sweep, the Runge-Kutta combination that follows it (``pdehip_jit_apply_stage``:
        kind 0 next stage input ``out2 = y + sum coefs*ks + c_new*k`` with k stored in ``k_out``; 1 RK4 update; 2 RKF45
        update + error norm into ``err``).  Returns False when the sweep is not available - then ``k_out`` holds the slope
        (plain ``apply``) and the caller combines with the pointwise kernels.  Two-pass chains keep their fused two-level
        sweep (tmp in registers) and combine separately."""
        if (not getattr(self, "_stage_ok", True) or self._pass_by_pass or self._fused_handle("scaled") is not None or self.has_reductions
                or self._reads_intermediate):
            self.apply(state, k_out, "scaled", dt, t)
            return False
        arrays = {"state": state, "out": k_out, **self.tmps, **self.aux}
        params = (C.c_double * 2)(dt, t)
        self._update_faces(t, state)
        last = len(self.plan.passes) - 1
        for i, p in enumerate(self.plan.passes):
            h, extras = self._kernel(i, "scaled")
            ex = (C.c_void_p * 3)()
            for m, name in enumerate(extras):
                ex[m] = arrays[name].ptr
            if i < last:
                self.lib.jit_apply(h, self.info.ref, arrays[p.src].ptr, ex, arrays[p.out].ptr, params, 2, self._faces(i), self.backend.stream)
                continue
            done = C.c_int(0)
            kp = (C.c_void_p * max(1, len(ks)))(*[k.ptr for k in ks])
            cf = (C.c_double * max(1, len(ks)))(*(list(coefs) if kind in (0, 4) else [0.0] * len(ks)))
            self.lib.jit_apply_stage(h, self.info.ref, arrays[p.src].ptr, ex, k_out.ptr, params, 2, self._faces(i), kind, y.ptr, len(ks), kp,
                                     cf, c_new, out2.ptr, err.ptr if err is not None else None, C.byref(done), self.backend.stream)
            if not done.value:
                self._stage_ok = False
                self.lib.jit_apply(h, self.info.ref, arrays[p.src].ptr, ex, k_out.ptr, params, 2, self._faces(i), self.backend.stream)
                return False
        return True

    # --- the whole fixed-step Euler loop in one C call (pdehip_jit_euler_run) ------------------------------------------------
    def loop_ok(self) -> bool:
        """The passes of this expression can run inside the C loops (``pdehip_jit_euler_run`` / ``pdehip_jit_rk_run``): no
        integrals (their values travel through the host) and no conditions given as Python functions (conditions that are
        expressions of time are refreshed on the device inside the loops: :meth:`bc_program`)."""
        # (decomposed grids - `_exchange` set -: only with the exchange descriptor the C loops read, `_exchange_desc`; the loops run the
        # passes one by one, which is all `_pass_by_pass` asks for)
        decomposed_ok = self._exchange is None or getattr(self, "_exchange_desc", None) is not None
        return (not self.has_reductions and not self._reads_intermediate and decomposed_ok
                and not any(getattr(tb, "host_only", False) for tb in self._dynamic))

    def bc_program(self):
        """Device program (``pde_hip.bc_expr.BcProgram``) of all time-dependent faces of this expression's tables, or None."""
        if not hasattr(self, "_bc_program"):
            from .bc_expr import program_for

            self._bc_program = program_for(self.lib, self._dynamic, self.info) if self._dynamic else None
        return self._bc_program

    def loop_passes(self, own: int, components: dict[str, int], fixed: list, keep: list, wrap: str = "euler", exchanged: set | None = None) -> list:
        """``pdehip_jit_pass_t`` entries of one evaluation of this equation (``wrap`` = "euler": one Euler step, the last pass
        writes ``state + dt*F``; "scaled": a Runge-Kutta slope, it writes ``dt*F``); ``own``: component of the state it
        advances, ``components``: the other variables' components by name, ``fixed``: list of device pointers that the
        entries index (extended here), ``keep``: objects that must outlive the descriptor."""
        index: dict[str, int] = {}

        def resolve(name: str, is_out: bool = False) -> int:
            if name == "state":
                return -1 - own
            if name == "out":
                assert is_out
                return -1 - own
            if name.startswith("var:"):
                return -1 - components[name[4:]]
            if name not in index:
                arr = self.tmps.get(name) or self.aux.get(name)
                index[name] = len(fixed)
                fixed.append(arr.ptr)
                keep.append(arr)
            return index[name]

        entries = []
        for i, p in enumerate(self.plan.passes):
            h, extras = self._kernel(i, wrap)
            e = _abi.JitPass()
            e.handle = h.value
            e.src = resolve(p.src)
            for m in range(3):
                e.extras[m] = resolve(extras[m]) if m < len(extras) else _abi.JIT_NONE
            e.out = resolve(p.out, True)
            faces = self._faces(i)
            e.faces = C.cast(faces, C.c_void_p).value if faces is not None else None
            keep.append(faces)
            # decomposed grids: the ghost layers of an operand travel before the FIRST pass of an evaluation that applies operators to
            # it (`exchanged`: what earlier passes - also of the other equations of a system - have exchanged already)
            desc = getattr(self, "_exchange_desc", None)
            if desc is not None and self.pass_faces[i] is not None and (exchanged is None or e.src not in exchanged):
                e.exchange = C.addressof(desc)
                keep.append(desc)
                if exchanged is not None:
                    exchanged.add(e.src)
            entries.append(e)
        return entries

    def euler_loop(self, state, other, dt: float, t0: float, nsteps: int):
        """``nsteps`` explicit Euler steps starting from ``state`` with ``other`` as the second buffer; returns the array that
        holds the result, or None when the loop is not available (then nothing was done)."""
        if not self.loop_ok():
            return None
        return _run_loop(self.lib, self.info, self._loop_desc("euler"), state, other, 1, dt, t0, nsteps, self.plan.uses_time, self.backend.stream,
                         self.bc_program())

    def _loop_desc(self, wrap: str):
        cache = self.__dict__.setdefault("_loops", {})
        if wrap not in cache:
            fixed: list = []
            keep: list = []
            entries = self.loop_passes(0, {}, fixed, keep, wrap, exchanged=set())
            cache[wrap] = ((_abi.JitPass * len(entries))(*entries), (C.c_void_p * max(1, len(fixed)))(*fixed), len(fixed), keep)
        return cache[wrap]

    def rk_run(self, y, ynew, work, err, dt: float, t0: float, nsteps: int, ctl=None, euler_adaptive: bool = False):
        """Runge-Kutta steps in ONE C call (``pdehip_jit_rk_run``): ``nsteps`` RK4 steps in place on ``y``, or - with ``ctl`` -
        the adaptive RKF45 loop (``euler_adaptive``: the reference's adaptive Euler loop, ``pdehip_jit_euler_adaptive_run``);
        returns the array holding the final state, or None when the loop is not available (integrals, function-valued
        conditions; two-pass chains of LARGE grids keep the Python loop, whose steps run the chain as one two-level sweep)."""
        if not self.loop_ok():
            return None
        chain = self._fused_handle("scaled") is not None
        if chain and int(np.prod(self.info.shape)) > (1 << 21):
            return None
        stage_fuse = getattr(self, "_stage_ok", True) and not chain and not self._pass_by_pass
        return _run_rk(self.lib, self.info, self._loop_desc("scaled"), 1, y, ynew, work, err, dt, t0, nsteps, ctl, stage_fuse, self.backend.stream,
                       self.bc_program(), euler_adaptive)

    def _fused_handle(self, wrap: str):
        if wrap not in self._fused:
            h = None
            ps = self.plan.passes
            if (not self.has_reductions and not self._reads_intermediate and not self._pass_by_pass and len(ps) == 2 and ps[0].src == "state" and not ps[0].extras
                    and ps[1].src == ps[0].out and ps[1].out == "out" and self.pass_faces[1] is not None):
                body1, ex1 = self.plan.epilogue(ps[0], "rate")
                body2, ex2 = self.plan.epilogue(ps[1], wrap)
                if not ex1 and ex2 in ([], ["state"]):
                    h = C.c_void_p()
                    self.lib.jit_create2(body1.encode(), body2.encode(), C.byref(h))
            self._fused[wrap] = h
        return self._fused[wrap]

    def _fused2(self, state, out, wrap: str, params) -> bool:
        """Two-pass chain ``tmp = f1(state)``, ``out = f2(tmp; state)`` in ONE sweep with tmp in registers (two-level
        kernel).  False when the plan has another shape or grid / BCs are not covered (then the passes run one by one)."""
        h = self._fused_handle(wrap)
        if h is None:
            return False
        done = C.c_int(0)
        faces_tmp = self.pass_faces[1]
        faces_u = self.pass_faces[0] or faces_tmp   # a first pass without operators never reads the halo of u
        self.lib.jit_fused2(h, self.info.ref, state.ptr, out.ptr, params, 2, faces_u.c, faces_tmp.c,
                            C.byref(done), self.backend.stream)
        if not done.value:
            self.lib.jit_destroy(h)
            self._fused[wrap] = None
        return bool(done.value)

    def euler2(self, state, out, dt: float) -> bool:
        """out = E(E(state)), E(u) = u + dt*F(u): TWO Euler steps in one sweep of the two-level kernel (intermediate level
        in registers).  Only for one-pass expressions of the state without extra arrays or explicit time; returns False
        (nothing done) when the expression, the grid or the BCs are not covered."""
        if self._two_ok is None:
            p = self.plan.passes[0]
            self._two_ok = (len(self.plan.passes) == 1 and p.src == "state" and not p.extras and not self.plan.uses_time
                            and self.pass_faces[0] is not None)
        if not self._two_ok:
            return False
        h, extras = self._kernel(0, "euler")
        if extras:
            self._two_ok = False
            return False
        params = (C.c_double * 2)(dt, 0.0)
        done = C.c_int(0)
        self.lib.jit_euler2(h, self.info.ref, state.ptr, out.ptr, params, 2, self.pass_faces[0].c, C.byref(done), self.backend.stream)
        if not done.value:
            self._two_ok = False
        return bool(done.value)

    def __del__(self):
        for h in getattr(self, "_fused", {}).values():
            if h is not None:
                try:
                    self.lib.jit_destroy(h)
                except Exception:  # noqa: BLE001 - interpreter shutdown
                    pass
        for h, _ in getattr(self, "_kernels", {}).values():
            try:
                self.lib.jit_destroy(h)
            except Exception:  # noqa: BLE001 - interpreter shutdown
                pass


class SystemRhs:
    """Right-hand side of a multi-field ``PDE({"u": ..., "v": ...})`` whose fields are all scalar (reference:
    ``pde/pdes/pde.py:401-499`` compiles one function per variable over a ``FieldCollection``): one
    :class:`ExpressionRhs` per equation, each seeing the other fields as centre-only inputs (or, where an operator is applied
    to another field, as the stencil array of a pass).  The state is ONE device array with a leading component axis, like
    ``FieldCollection.data``.  Same evaluator interface as :class:`ExpressionRhs`; the fused sweeps of single-field
    expressions (two steps per sweep, stage epilogues) do not apply — stages combine with the pointwise kernels."""

    def __init__(self, variables: list[str], parts: list["ExpressionRhs"], info):
        self.variables, self.parts, self.info = list(variables), list(parts), info
        self.ncomp = len(self.variables)
        # conditions that read the field: an operator of one equation may act on ANOTHER field of the system - refreshed pass by
        # pass from the array the pass reads (the component views below), which keeps the system out of the C loops
        for part in self.parts:
            if any(getattr(tb, "reads_value", False) for tb in part._dynamic):
                part._reads_intermediate = True

    def apply(self, state, out, wrap: str = "rate", dt: float = 0.0, t: float = 0.0) -> None:
        state, out = state.flat(), out.flat()   # (a lone rank-2 field arrives with two tensor axes)
        comps = {name: state.component(k) for k, name in enumerate(self.variables)}
        for k, (name, part) in enumerate(zip(self.variables, self.parts)):
            others = {n: a for n, a in comps.items() if n != name}
            part.apply(comps[name], out.component(k), wrap, dt, t, others=others)

    def apply_stage(self, state, k_out, dt, t, kind, y, ks, coefs, c_new, out2, err=None) -> bool:
        self.apply(state, k_out, "scaled", dt, t)
        return False

    def euler2(self, state, out, dt: float) -> bool:
        return False

    def euler_loop(self, state, other, dt: float, t0: float, nsteps: int):
        """The fixed-step Euler loop of the whole system in one C call (every equation reads the current state of all
        fields and writes its component of the next one); None when an equation cannot take part."""
        if not all(p.loop_ok() for p in self.parts):
            return None
        part0 = self.parts[0]
        uses_time = any(p.plan.uses_time for p in self.parts)
        return _run_loop(part0.lib, self.info, self._loop_desc("euler"), state, other, self.ncomp, dt, t0, nsteps, uses_time, part0.backend.stream,
                         self.bc_program())

    def _loop_desc(self, wrap: str):
        cache = self.__dict__.setdefault("_loops", {})
        if wrap not in cache:
            fixed: list = []
            keep: list = []
            components = {name: k for k, name in enumerate(self.variables)}
            entries = []
            exchanged: set = set()      # (decomposed grids: an operand travels once per evaluation of the whole system)
            for k, part in enumerate(self.parts):
                entries += part.loop_passes(k, components, fixed, keep, wrap, exchanged=exchanged)
            cache[wrap] = ((_abi.JitPass * len(entries))(*entries), (C.c_void_p * max(1, len(fixed)))(*fixed), len(fixed), keep)
        return cache[wrap]

    def bc_program(self):
        """ONE device program for the time-dependent faces of all equations' tables (or None)."""
        if not hasattr(self, "_bc_program"):
            from .bc_expr import program_for

            tables = [tb for part in self.parts for tb in part._dynamic]
            self._bc_program = program_for(self.parts[0].lib, tables) if tables else None
        return self._bc_program

    def rk_run(self, y, ynew, work, err, dt: float, t0: float, nsteps: int, ctl=None, euler_adaptive: bool = False):
        """Runge-Kutta steps of the whole system in one C call (see :meth:`ExpressionRhs.rk_run`): every equation evaluates its
        component of the slope from the stage input of all fields; the combinations run on all components at once."""
        if not all(p.loop_ok() for p in self.parts):
            return None
        part0 = self.parts[0]
        # (stage_fuse bit 1: the components are the (re, im) pairs of complex fields - modulus error norm from an explicit error field,
        # the last array of `work`)
        pairs = 2 if (ctl is not None and getattr(self, "complex_pairs", False)) else 0
        return _run_rk(part0.lib, self.info, self._loop_desc("scaled"), self.ncomp, y, ynew, work, err, dt, t0, nsteps, ctl, pairs,
                       part0.backend.stream, self.bc_program(), euler_adaptive)
