"""Complex-valued expression PDEs as real systems ("fp64 / fp32 / complex").

The reference specialises its operators and right-hand sides for complex arrays (numba type specialisation,
``pde/backends/numba/operators/cartesian.py``; the controller turns the state complex when ``pde.complex_valued``,
``pde/solvers/controller.py:430-432``; expressions may contain ``I``, ``pde/pdes/pde.py:299-399``).  Every stencil of this path has
REAL coefficients, so a complex field ``u = a + i b`` is carried on the device as the two real components ``(a, b)`` - planar, like the
components of a vector field - and an equation ``du/dt = F(u)`` becomes the real system ``da/dt = Re F``, ``db/dt = Im F``:

* linear operators act on the parts separately, ``laplace(a + i b) = laplace(a) + i laplace(b)`` (the same for ``d_dx``, ``d2_dx2`` ...);
  ``gradient_squared(w) = sum (d w)^2`` gives ``gs(a) - gs(b) + 2 i sum d(a) d(b)``;
* the vector operators (round 5) are lowered to their per-axis atoms first (:func:`lower_vector_operators`: ``gradient(w)`` ->
  ``[grad_k(w)]``, ``divergence(v)`` -> ``sum div_k(v_k)``, ``vector_laplace(v)`` -> ``[vlap_k(v_k)]``, ``dot(v, w)`` ->
  ``sum v_k conjugate(w_k)`` like ``make_dot_operator`` of the reference, pde/fields/datafield_base.py:965-986), which are linear with
  real coefficients again;
* everything pointwise (``I``, products, integer powers, ``conjugate``, ``Abs``, ``re``, ``im``, ``exp``) is split by sympy's
  ``as_real_imag`` after the operator applications have been replaced by real place-holders.

The split is symbolic and happens once per equation; the kernels then see ordinary real expressions (``pde_hip/expr.py``).  Steppers act
componentwise, which is exactly complex arithmetic with real step sizes; only the error norm of the adaptive schemes is the complex
modulus (``np.abs`` of a complex array, ``pde/solvers/runge_kutta.py:147-148``): ``pdehip_max_abs_pairs``.
"""

from __future__ import annotations

from typing import Any

IM_OPERAND = "_imop"     # suffix of an operator name whose operand is the IMAGINARY part of the operator's complex argument
# Conditions with a COMPLEX factor (mixed conditions with a complex coefficient) couple the parts: Re ghost = Re c + Re f * a - Im f * b,
# Im ghost = Im c + Re f * b + Im f * a for the operand a + i b.  The stencils are linear in their ghost cells, so the coupling terms are
# differences of two applications of the same stencil to the OTHER part: one with the conditions (0, -/+ Im f), one with (0, 0) - the
# interiors cancel exactly, what remains is the contribution of -/+ Im f * (adjacent cell) at the cells next to those faces.  Suffixes of
# the operator names -> ``convert_bcs(part=...)``:
COUPLING_SUFFIXES = {"_cplre": "cpl-", "_cplim": "cpl+", "_cplz": "zero"}


def part_names(var: str) -> tuple[str, str]:
    """Names of the real and the imaginary part of ``var`` inside the real system."""
    return f"{var}_re_", f"{var}_im_"


def lower_vector_operators(e, nd: int):
    """Vector-valued sub-expressions of a COMPLEX expression component by component (the twin of ``ExpressionPlan._lower_vectors`` for
    scalar complex fields: there are no vector fields in a complex state, vectors only arise from ``gradient``): returns ``("s", expr)``
    or ``("v", [components])`` with the atoms ``grad_<k>`` / ``div_<k>`` / ``vlap_<k>`` (one scalar argument each)."""
    import sympy as sp

    def fn(name):
        return sp.Function(name)

    if isinstance(e, sp.core.function.AppliedUndef):
        name = e.func.__name__
        args = [lower_vector_operators(a, nd) for a in e.args]
        if name == "gradient":
            if len(args) != 1 or args[0][0] != "s":
                msg = "hip backend: `gradient` inside expressions takes one scalar argument"
                raise NotImplementedError(msg)
            return "v", [fn(f"grad_{k}")(args[0][1]) for k in range(nd)]
        if name == "divergence":
            if len(args) != 1 or args[0][0] != "v":
                msg = "`divergence` needs a vector argument"
                raise ValueError(msg)
            return "s", sp.Add(*[fn(f"div_{k}")(c) for k, c in enumerate(args[0][1])])
        if name == "vector_laplace":
            if len(args) != 1 or args[0][0] != "v":
                msg = "`vector_laplace` needs a vector argument"
                raise ValueError(msg)
            return "v", [fn(f"vlap_{k}")(c) for k, c in enumerate(args[0][1])]
        if name in ("dot", "inner"):
            if len(args) != 2 or args[0][0] != "v" or args[1][0] != "v":
                msg = "hip backend: `dot` inside complex-valued expressions takes two vectors"
                raise NotImplementedError(msg)
            return "s", sp.Add(*[x * sp.conjugate(y) for x, y in zip(args[0][1], args[1][1])])
        if name in ("vector_gradient", "tensor_divergence", "outer"):
            msg = f"hip backend: operator `{name}` inside complex-valued expressions is not supported"
            raise NotImplementedError(msg)
        if any(k != "s" for k, _ in args):
            msg = f"hip backend: operator `{name}` of a vector inside expressions is not supported"
            raise NotImplementedError(msg)
        return "s", e.func(*[a for _, a in args])
    if not e.args:
        return "s", e
    parts = [lower_vector_operators(a, nd) for a in e.args]
    if all(k == "s" for k, _ in parts):
        return "s", e.func(*[a for _, a in parts])
    if e.is_Add:
        if any(k != "v" for k, _ in parts):
            msg = "cannot add fields of different rank"
            raise ValueError(msg)
        return "v", [sp.Add(*[p[1][k] for p in parts]) for k in range(nd)]
    if e.is_Mul and sum(k != "s" for k, _ in parts) == 1:
        val = next(p[1] for p in parts if p[0] != "s")
        scal = sp.Mul(*[p[1] for p in parts if p[0] == "s"])
        return "v", [scal * c for c in val]
    msg = f"hip backend: vector expression `{e}` is not supported (sums, scalar multiples, dot, divergence)"
    raise NotImplementedError(msg)


def split_expression(expr_str: str, variables: list[str], consts: dict[str, Any], axes: tuple[str, ...], aliases: dict[str, str] | None = None,
                     user_funcs: dict[str, Any] | None = None, coupled=None) -> tuple[str, str, dict[str, float], dict[str, str]]:
    """Real and imaginary part of the right-hand side ``expr_str`` as expression strings over the parts ``part_names(v)`` of the
    (complex) fields ``variables``; complex scalar constants are folded in, real ones stay symbols.  Returns ``(re, im, consts, aliases)``:
    the constants the new expressions still need and the operator aliases they use - ``laplace(w)`` of a complex argument ``w`` becomes
    ``laplace(Re w) + I * laplace_imop(Im w)``: the same stencil, but the conditions of the second operand are the IMAGINARY parts of the
    operator's boundary values (``convert_bcs(part="im")``), whatever equation the term ends up in.  ``coupled(base) -> bool``: the
    conditions of operator ``base`` have complex factors - its applications get the coupling terms (COUPLING_SUFFIXES)."""
    import sympy as sp

    if user_funcs:
        msg = "hip backend: user functions inside complex-valued expressions are not supported"
        raise NotImplementedError(msg)
    aliases = aliases or {}
    linear_axis_ops = {f"d_d{ax}" for ax in axes} | {f"d2_d{ax}2" for ax in axes}
    expr_str = expr_str.replace("∇²", "laplace").replace("^", "**")
    local: dict[str, Any] = {"I": sp.I}
    fields = {v: sp.Symbol(v) for v in variables}
    local.update(fields)
    real_syms: dict[str, Any] = {}
    keep: dict[str, float] = {}
    folded: dict[Any, Any] = {}
    for name, value in consts.items():
        if name in fields:
            continue
        sym = sp.Symbol(name, real=True)
        real_syms[name] = sym
        local[name] = sym
        try:
            scalar = complex(value)
        except (TypeError, ValueError):
            import numpy as np

            if np.iscomplexobj(getattr(value, "data", value)):     # (its imaginary part would be dropped silently)
                msg = f"hip backend: the array-valued constant `{name}` of a complex-valued expression must be real"
                raise NotImplementedError(msg)
            keep[name] = value          # an array / field on the grid: stays a (real) symbol of the plan
            continue
        if scalar.imag != 0:
            folded[sym] = sp.Float(scalar.real, 17) + sp.I * sp.Float(scalar.imag, 17)
        else:
            keep[name] = scalar.real
    for name in (*axes, "t"):
        if name not in local:
            local[name] = sp.Symbol(name, real=True)
    # every other name followed by "(" is an operator: an undefined function
    import re as _re

    for name in set(_re.findall(r"([A-Za-z_][A-Za-z_0-9]*)\s*\(", expr_str)):
        if name not in local and not hasattr(sp, name):
            local[name] = sp.Function(name)
    for name in ("laplace", "gradient_squared", *linear_axis_ops, *aliases):
        local.setdefault(name, sp.Function(name))
    expr = sp.sympify(expr_str, locals=local)
    nd = len(axes)
    vector_atoms = {f"{kind}_{k}" for kind in ("grad", "div", "vlap") for k in range(nd)}
    if any(f.func.__name__ in ("gradient", "divergence", "vector_laplace", "dot", "inner", "vector_gradient", "tensor_divergence", "outer")
           for f in expr.atoms(sp.core.function.AppliedUndef)):
        kind, expr = lower_vector_operators(expr, nd)
        if kind != "s":
            msg = f"hip backend: the right-hand side `{expr_str}` is a vector, the field is a scalar"
            raise NotImplementedError(msg)
    parts = {fields[v]: tuple(sp.Symbol(n, real=True) for n in part_names(v)) for v in variables}
    holders: dict[Any, Any] = {}
    new_aliases: dict[str, str] = dict(aliases)

    def hold(call) -> Any:
        sym = sp.Symbol(f"_op{len(holders)}_", real=True)
        holders[sym] = call
        return sym

    def split(e) -> tuple[Any, Any]:
        """(real part, imaginary part) of ``e`` as expressions over the real symbols and the place-holders."""
        undefined = [a for a in e.atoms(sp.core.function.AppliedUndef)]
        # innermost first: replace operator applications by (holder_re + I holder_im)
        sub: dict[Any, Any] = {}
        for call in sorted(undefined, key=lambda c: len(str(c))):
            if call in sub:
                continue
            name = call.func.__name__
            base = aliases.get(name, name)
            if len(call.args) != 1:
                msg = f"hip backend: operator `{name}` with {len(call.args)} arguments inside a complex-valued expression"
                raise NotImplementedError(msg)
            ar, ai = split(call.args[0].xreplace(sub))
            fn = call.func
            if base == "laplace" or base in linear_axis_ops or base in vector_atoms:
                # linear with real coefficients (the Laplacian, d_dx, d2_dx2 ..., the per-axis atoms of the vector operators): acts on the
                # parts separately; the imaginary operand takes the imaginary parts of the operator's boundary values
                re_part = hold(fn(ar)) if ar != 0 else sp.Integer(0)
                im_part = sp.Integer(0)
                if ai != 0:
                    new_aliases[name + IM_OPERAND] = base
                    im_part = hold(sp.Function(name + IM_OPERAND)(ai))
                if coupled is not None and coupled(base):
                    if base in vector_atoms:
                        msg = f"hip backend: conditions with complex factors on the vector operator behind `{name}` are not supported"
                        raise NotImplementedError(msg)
                    for sfx in COUPLING_SUFFIXES:
                        new_aliases[name + sfx] = base
                    if ai != 0:
                        re_part = re_part + hold(sp.Function(name + "_cplre")(ai)) - hold(sp.Function(name + "_cplz")(ai))
                    if ar != 0:
                        im_part = im_part + hold(sp.Function(name + "_cplim")(ar)) - hold(sp.Function(name + "_cplz")(ar))
            elif base == "gradient_squared" and coupled is not None and coupled(base):
                msg = "hip backend: `gradient_squared` of a complex field with complex-factor conditions is not supported"
                raise NotImplementedError(msg)
            elif base == "gradient_squared" and ai == 0:
                re_part, im_part = hold(fn(ar)), sp.Integer(0)
            elif base == "gradient_squared" and name == base:
                # sum_axes (d w)^2 of w = a + i b (no conjugate: cartesian.py:661-668 squares the complex central differences) is
                # gs(a) - gs(b) + 2 i sum_axes d(a) d(b); the per-axis central differences read the ghost cells of THIS operator's
                # conditions: aliases of d_d<ax> named after it (`gradient_squared_d<ax>`, `..._imop` for the imaginary operand)
                re_part = (hold(fn(ar)) if ar != 0 else sp.Integer(0)) - hold(sp.Function(name + IM_OPERAND)(ai))
                new_aliases[name + IM_OPERAND] = base
                im_part = sp.Integer(0)
                if ar != 0:
                    for ax in axes:
                        d_re, d_im = f"{name}_d{ax}", f"{name}_d{ax}{IM_OPERAND}"
                        new_aliases[d_re] = new_aliases[d_im] = f"d_d{ax}"
                        im_part = im_part + 2 * hold(sp.Function(d_re)(ar)) * hold(sp.Function(d_im)(ai))
            else:
                # (the vector operators would split like laplace: not built)
                msg = f"hip backend: operator `{name}` of a complex argument inside an expression is not supported"
                raise NotImplementedError(msg)
            sub[call] = re_part + sp.I * im_part
        e = e.xreplace(sub).xreplace(folded)
        e = e.xreplace({f: p[0] + sp.I * p[1] for f, p in parts.items()})
        re_part, im_part = sp.expand(e).as_real_imag()
        leftovers = (re_part.atoms(sp.re, sp.im, sp.arg) | im_part.atoms(sp.re, sp.im, sp.arg))
        if leftovers:
            msg = f"hip backend: cannot split `{e}` into real and imaginary part ({sorted(map(str, leftovers))[:3]})"
            raise NotImplementedError(msg)
        return re_part, im_part

    re_part, im_part = split(expr)
    # put the operator applications back (outermost holders contain inner ones)
    for _ in range(len(holders) + 1):
        if not (re_part.free_symbols | im_part.free_symbols) & set(holders):
            break
        re_part, im_part = re_part.xreplace(holders), im_part.xreplace(holders)
    return sp.sstr(re_part), sp.sstr(im_part), keep, new_aliases
