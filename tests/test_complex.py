"""Complex fields (SURVEY 8 row a1 "fp64 / fp32 / complex"; VERDICT r3 "missing #2"): ``pde.complex_valued`` equations - expressions with
``I``, complex constants, complex boundary values - and operators on complex data.

Every stencil of the path has real coefficients: a complex field travels as planar (re, im) pairs of the real type and its equation as the
real system of the parts (``pde_hip/complex_expr.py``).  CPU part: the symbolic split against sympy / numpy directly, the oracle's modulus
norm, and the product's host side through the REAL py-pde (host shim) against the reference's numpy solver (with the right-hand side
written with its field operators: ``pde.PDE`` needs numba there) and its torch backend.  GPU part: tests/test_hip_complex.py.
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest
from refpath import REF  # noqa: E402


def test_symbolic_split_against_complex_arithmetic():
    """Re / Im of pointwise expressions == numpy's complex arithmetic on random data."""
    import sympy as sp

    from pde_hip.complex_expr import split_expression

    rng = np.random.default_rng(0)
    a, b = rng.uniform(-1, 1, 50), rng.uniform(-1, 1, 50)
    u = a + 1j * b
    for expr, ref in (("(1 + 2*I) * u * Abs(u)**2 - I * u**3", (1 + 2j) * u * np.abs(u) ** 2 - 1j * u**3),
                      ("conjugate(u) * u + g * u", np.conj(u) * u + (0.5 - 0.25j) * u),
                      ("exp(I * t) * u + re(u) - 2 * im(u)", np.exp(0.3j) * u + u.real - 2 * u.imag)):
        re_s, im_s, consts, aliases = split_expression(expr, ["u"], {"g": 0.5 - 0.25j}, ("x",))
        syms = {"u_re_": a, "u_im_": b, "t": 0.3, **consts}
        f = [sp.lambdify(list(map(sp.Symbol, syms)), sp.sympify(s), modules="numpy")(*syms.values()) for s in (re_s, im_s)]
        np.testing.assert_allclose(f[0] + 1j * f[1], ref, rtol=1e-13, atol=1e-14)
    re_s, im_s, _, aliases = split_expression("I * laplace(u**2) + laplace(laplace(u))", ["u"], {}, ("x",))
    assert "laplace_imop" in aliases and aliases["laplace_imop"] == "laplace"
    # Re [i lap(u^2)] = -lap(Im u^2) = -lap(2ab) through the IMAGINARY-operand operator; Re lap lap u = lap lap a
    assert "laplace_imop(2*u_im_*u_re_)" in re_s.replace(" ", "").replace("2*u_re_*u_im_", "2*u_im_*u_re_") and "laplace(laplace(u_re_))" in re_s
    # per-axis derivatives split like the Laplacian; gradient_squared(a + i b) = gs(a) - gs(b) + 2 i sum_axes d(a) d(b) with central differences
    # that carry the operator's own conditions (aliases of d_d<ax> named after it)
    re_s, im_s, _, aliases = split_expression("d_dx(u) + gradient_squared(u)", ["u"], {}, ("x", "y"))
    assert aliases["d_dx_imop"] == "d_dx" and aliases["gradient_squared_imop"] == "gradient_squared"
    assert aliases["gradient_squared_dy"] == "d_dy" and aliases["gradient_squared_dx_imop"] == "d_dx"
    flat = lambda text: text.replace(" ", "")      # noqa: E731
    assert "d_dx(u_re_)" in flat(re_s) and "gradient_squared(u_re_)" in flat(re_s) and "-gradient_squared_imop(u_im_)" in flat(re_s)
    assert "d_dx_imop(u_im_)" in flat(im_s) and "2*gradient_squared_dx(u_re_)*gradient_squared_dx_imop(u_im_)" in flat(im_s)
    # vector operators (round 5): lowered to per-axis atoms that are linear with real coefficients; `dot` conjugates its second operand
    re_s, im_s, _, aliases = split_expression("divergence(gradient(u)) + dot(gradient(u), gradient(u))", ["u"], {}, ("x", "y"))
    assert aliases["grad_0_imop"] == "grad_0" and aliases["div_1_imop"] == "div_1"
    assert "div_0(grad_0(u_re_))" in flat(re_s) and "div_1_imop(grad_1_imop(u_im_))" in flat(im_s)
    assert "grad_0(u_re_)**2" in flat(re_s) and "grad_1_imop(u_im_)**2" in flat(re_s) and "grad" not in flat(im_s).replace("div_0_imop(grad_0_imop(", "").replace("div_1_imop(grad_1_imop(", "")
    with pytest.raises(NotImplementedError):
        split_expression("tensor_divergence(outer(gradient(u), gradient(u)))", ["u"], {}, ("x", "y"))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_oracle_modulus_norm(dtype):
    from helpers import oracle_grid, to_full

    import pde_hip
    from oracle import pde_oracle as O

    rng = np.random.default_rng(1)
    grid = pde_hip.UnitGrid([5, 6, 7])
    z = (rng.normal(size=(2, *grid.shape)) + 1j * rng.normal(size=(2, *grid.shape))).astype(np.complex128 if dtype == np.float64 else np.complex64)
    planar = np.stack([z.real, z.imag], axis=1).reshape(4, *grid.shape).astype(dtype)
    got = O.max_abs_pairs(oracle_grid(grid, dtype), 2, to_full(grid, planar))
    assert got == pytest.approx(float(np.abs(z).max()), rel=1e-15 if dtype == np.float64 else 1e-6)
    planar[3, 1, 2, 3] = np.nan
    assert np.isnan(O.max_abs_pairs(oracle_grid(grid, dtype), 2, to_full(grid, planar)))


if not (REF / "pde").exists():
    pytest.skip("py-pde (reference) not available", allow_module_level=True)
if str(REF) not in sys.path:
    sys.path.append(str(REF))

import pde  # noqa: E402
import shimlib  # noqa: E402
from helpers import max_rel  # noqa: E402


@pytest.fixture(autouse=True)
def _scipy_operators(monkeypatch):
    monkeypatch.setitem(pde.config, "default_backend", "scipy")
    monkeypatch.setitem(pde.config, "backend.torch.compile", False)


@pytest.fixture(params=[False, True], ids=["unfused", "fused"])
def hip(request):
    with shimlib.use_shim(fused=request.param):
        import pde_hip.pypde_plugin  # noqa: F401

        yield pde.backends.get_backend("hip")


_BC = {"x": {"value": 1 + 2j}, "y": "periodic"}


class _Schroedinger(pde.PDEBase):
    """`PDE({'p': 'I * laplace(p)'})` for the reference's numpy solver (pde.PDE takes its operators from numba there)."""

    complex_valued = True

    def evolution_rate(self, state, t=0):
        return 1j * state.laplace(_BC)


@pytest.mark.parametrize("solver,adaptive", [("euler", False), ("runge-kutta", False), ("runge-kutta", True), ("euler", True), ("adams-bashforth", False)])
def test_schroedinger_against_the_reference(hip, solver, adaptive):
    """VERDICT r3 "next #5": `PDE({'u': 'I*laplace(u)'})` with a complex boundary value, Euler / RK4 / RKF45 / adaptive Euler / AB2:
    equal step counts, <= 1e-10 of the reference's numpy backend; a REAL initial field is turned complex by the controller."""
    grid = pde.UnitGrid([8, 6], periodic=[False, True])
    rng = np.random.default_rng(0)
    for field in (pde.ScalarField(grid, rng.uniform(-1, 1, grid.shape) + 1j * rng.uniform(-1, 1, grid.shape)), pde.ScalarField.random_uniform(grid, rng=rng)):
        kw = dict(t_range=0.05, dt=1e-3, solver=solver, tracker=None, ret_info=True)
        if adaptive:
            kw["adaptive"] = True
        ref, iref = _Schroedinger().solve(field, backend="numpy", **kw)
        res, info = pde.PDE({"p": "I * laplace(p)"}, bc=_BC).solve(field, backend="hip", **kw)
        got = np.array(res.data)
        assert got.dtype == np.complex128 and res.is_complex
        assert info["solver"]["steps"] == iref["solver"]["steps"]
        assert max_rel(got, ref.data) < 1e-10


@pytest.mark.parametrize("solver,adaptive", [("euler", False), ("runge-kutta", True)])
def test_nonlinear_complex_equation(hip, solver, adaptive):
    """Gross-Pitaevskii-like right-hand side with a complex coefficient, `Abs`, `conjugate` and a complex Neumann value; yardstick: the
    reference's numpy solver around the same right-hand side written with its field operators."""
    grid = pde.CartesianGrid([[0, 10], [0, 8]], [10, 8], periodic=[True, False])
    rng = np.random.default_rng(2)
    field = pde.ScalarField(grid, rng.uniform(-0.5, 0.5, grid.shape) + 1j * rng.uniform(-0.5, 0.5, grid.shape))
    bc = {"x": "periodic", "y": {"derivative": 0.1 - 0.2j}}
    g = 0.3 - 0.8j

    class Restated(pde.PDEBase):
        complex_valued = True

        def evolution_rate(self, state, t=0):
            c = state.data
            return pde.ScalarField(state.grid, -1j * state.laplace(bc).data + g * c * np.abs(c) ** 2 - 0.1 * np.conjugate(c))

    eq = pde.PDE({"c": "-I * laplace(c) + g * c * Abs(c)**2 - 0.1 * conjugate(c)"}, consts={"g": g}, bc=bc)
    kw = dict(t_range=0.05, dt=1e-3, solver=solver, tracker=None, ret_info=True, adaptive=adaptive)
    ref, iref = Restated().solve(field, backend="numpy", **kw)
    res, info = eq.solve(field, backend="hip", **kw)
    assert info["solver"]["steps"] == iref["solver"]["steps"]
    assert max_rel(np.array(res.data), ref.data) < 1e-10


def test_axis_derivatives_and_gradient_squared_of_complex_fields(hip):
    """`d_dx`, `d2_dy2` and `gradient_squared` of a complex field inside an expression (complex boundary values, one periodic axis) against the
    reference's numpy solver around the same right-hand side written with its field operators."""
    grid = pde.CartesianGrid([[0, 5], [0, 4]], [10, 8], periodic=[True, False])
    rng = np.random.default_rng(7)
    field = pde.ScalarField(grid, rng.uniform(-0.5, 0.5, grid.shape) + 1j * rng.uniform(-0.5, 0.5, grid.shape))
    bc = {"x": "periodic", "y": {"value": 0.2 + 0.1j}}

    class Restated(pde.PDEBase):
        complex_valued = True

        def evolution_rate(self, state, t=0):
            grad = state.gradient(bc).data
            work = state.copy()
            work.set_ghost_cells(bc)
            full = work._data_full      # d2_dy2 of the reference (operators/common.py:150-190) by hand: its scipy backend has no such operator
            d2y = (full[1:-1, 2:] - 2 * full[1:-1, 1:-1] + full[1:-1, :-2]) / state.grid.discretization[1] ** 2
            if self.which == "gs":
                return pde.ScalarField(state.grid, (0.5 + 0.2j) * (grad[0] ** 2 + grad[1] ** 2) - 0.1 * state.data)
            return pde.ScalarField(state.grid, -1j * grad[0] + 0.3 * d2y - 0.1 * state.data)

    for which, expr in (("gs", "(0.5 + 0.2*I) * gradient_squared(c) - 0.1 * c"), ("axis", "-I * d_dx(c) + 0.3 * d2_dy2(c) - 0.1 * c")):
        eq, restated = pde.PDE({"c": expr}, bc=bc), Restated()
        restated.which = which
        for solver in ("euler", "runge-kutta"):
            kw = dict(t_range=0.03, dt=1e-3, solver=solver, tracker=None)
            ref = restated.solve(field, backend="numpy", **kw)
            res = eq.solve(field, backend="hip", **kw)
            assert max_rel(np.array(res.data), ref.data) < 1e-10, (which, solver)


def test_two_complex_fields_and_a_class_pde(hip):
    """A collection of two complex scalar fields (coupled), and DiffusionPDE on a complex state (linear: acts on the parts)."""
    grid = pde.UnitGrid([9, 7], periodic=[True, False])
    rng = np.random.default_rng(3)
    fields = [pde.ScalarField(grid, rng.uniform(-1, 1, grid.shape) + 1j * rng.uniform(-1, 1, grid.shape), label=n) for n in "ab"]
    state = pde.FieldCollection(fields)
    eq = pde.PDE({"a": "I * laplace(a) - b", "b": "laplace(b) + I * a"}, bc={"x": "periodic", "y": {"value": 0.5j}})
    ref = eq.solve(state, t_range=0.02, dt=1e-3, backend="torch", tracker=None)
    res = eq.solve(state, t_range=0.02, dt=1e-3, backend="hip", tracker=None)
    assert max_rel(np.array(res.data), ref.data) < 1e-10
    diff = pde.DiffusionPDE(0.7, bc={"x": "periodic", "y": {"value": 1 - 1j}})
    for solver in ("euler", "runge-kutta"):
        ref = diff.solve(fields[0], t_range=0.1, dt=0.01, solver=solver, backend="numpy", tracker=None)
        res = diff.solve(fields[0], t_range=0.1, dt=0.01, solver=solver, backend="hip", tracker=None)
        assert max_rel(np.array(res.data), ref.data) < 1e-10


def test_operators_on_complex_fields(hip):
    """`field.laplace(bc, backend="hip")`, `gradient`, `grid.make_operator(...)` on complex data: the parts through the real kernels."""
    grid = pde.CartesianGrid([[0, 4], [0, 4.5]], [8, 9], periodic=[False, True])   # (dx = 0.5: the scipy yardstick wants one spacing)
    rng = np.random.default_rng(4)
    field = pde.ScalarField(grid, rng.uniform(-1, 1, grid.shape) + 1j * rng.uniform(-1, 1, grid.shape))
    bc = {"x": {"value": 0.3 - 0.7j}, "y": "periodic"}
    np.testing.assert_allclose(field.laplace(bc, backend="hip").data, field.laplace(bc, backend="scipy").data, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(field.gradient(bc, backend="hip").data, field.gradient(bc, backend="scipy").data, rtol=1e-12, atol=1e-12)
    op = grid.make_operator("laplace", bc=bc, backend="hip", dtype=complex)
    np.testing.assert_allclose(op(field.data), field.laplace(bc, backend="scipy").data, rtol=1e-12, atol=1e-12)
    with pytest.raises(NotImplementedError):
        grid.make_operator("gradient_squared", bc=bc, backend="hip", dtype=complex)


def test_expression_conditions_on_complex_fields(hip):
    """Round 5 (VERDICT r4 "missing #4"): conditions given as expressions / Python functions of time and position with COMPLEX values on a
    complex field - operators and all four steppers against the reference (its scipy operators; right-hand side restated with its field
    operators).  `F = A + B * value` with a real `B` acts on the parts separately; a complex `B` or an `F` that is not affine in `value`
    couples the parts and is refused."""
    grid = pde.CartesianGrid([[0, 2], [0, 1.5]], [8, 6], periodic=[False, True])
    rng = np.random.default_rng(5)
    field = pde.ScalarField(grid, rng.normal(size=grid.shape) + 1j * rng.normal(size=grid.shape))
    bc = {"x-": {"value_expression": "I*t + y"}, "x+": {"derivative_expression": "(1 + 2*I)*cos(t) - 0.5*value"}, "y": "periodic"}
    bc_func = {"x-": {"value_expression": lambda v, dx, x, y, t: 1j * t + y}, "x+": {"derivative": 0.2 - 0.1j}, "y": "periodic"}
    for conditions in (bc, bc_func):
        for name in ("laplace", "gradient"):
            ref = grid.make_operator(name, bc=conditions, backend="scipy", dtype=complex)(field.data, args={"t": 0.7})
            got = grid.make_operator(name, bc=conditions, backend="hip", dtype=complex)(field.data, args={"t": 0.7})
            np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)

    parsed = grid.get_boundary_conditions(bc)      # (once: the reference parses the expressions of a dict at every call)

    class Restated(pde.PDEBase):
        complex_valued = True

        def evolution_rate(self, state, t=0):
            c = state.data
            return pde.ScalarField(state.grid, 1j * state.laplace(parsed, args={"t": t}).data - 0.1 * c * np.abs(c) ** 2)

    eq = pde.PDE({"u": "I * laplace(u) - 0.1 * u * Abs(u)**2"}, bc=bc)
    for solver, adaptive in (("euler", False), ("runge-kutta", False), ("runge-kutta", True), ("euler", True)):
        kw = dict(t_range=0.02, dt=1e-3, solver=solver, adaptive=adaptive, tracker=None, ret_info=True)
        ref, iref = Restated().solve(field, backend="numpy", **kw)
        res, info = eq.solve(field, backend="hip", **kw)
        assert info["solver"]["steps"] == iref["solver"]["steps"]
        assert max_rel(np.array(res.data), ref.data) < 1e-10, (solver, adaptive)
    # a complex slope couples the parts: supported for expressions since round 6 (two more stencil applications per part) ...
    slope = {"x-": {"derivative_expression": "(1 + 2*I) * value + cos(t) * y"}, "x+": {"value": 0.3j}, "y": "periodic"}
    for name in ("laplace", "gradient"):
        ref = grid.make_operator(name, bc=slope, backend="scipy", dtype=complex)(field.data, args={"t": 0.7})
        got = grid.make_operator(name, bc=slope, backend="hip", dtype=complex)(field.data, args={"t": 0.7})
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    # ... refused: conditions that are not affine in the value, Python functions with a complex slope
    for coupled in ({"value_expression": "value**2"}, {"value_expression": lambda v, dx, x, y, t: 1j * v}):
        with pytest.raises(NotImplementedError, match="couples real and imaginary part"):
            grid.make_operator("laplace", bc={"x-": coupled, "x+": {"value": 0}, "y": "periodic"}, backend="hip", dtype=complex)(field.data, args={"t": 0.0})


def test_vector_operators_of_complex_arguments(hip):
    """Round 5 (VERDICT r4 "missing #4"): `gradient` / `divergence` / `vector_laplace` / `dot` of complex arguments inside an expression
    against the reference's torch backend (Euler; it has no Runge-Kutta) and, for the adaptive schemes, against its numpy solver around the
    same right-hand side written with its field operators.  `dot` conjugates its second operand (datafield_base.py:965-986)."""
    grid = pde.CartesianGrid([[0, 2], [0, 1.5]], [8, 6], periodic=[True, False])
    rng = np.random.default_rng(6)
    field = pde.ScalarField(grid, rng.normal(size=grid.shape) + 1j * rng.normal(size=grid.shape))
    bc = {"x": "periodic", "y": {"value": 0.3 - 0.2j}}
    first = "I * divergence((1 + 0.5*I) * gradient(c)) + 0.1 * dot(gradient(c), gradient(c)) - 0.1 * c"
    for rhs in (first, "divergence(gradient(c)) + I * c", "dot(gradient(c), gradient(c**2)) * I - inner(I * gradient(c), gradient(c))",
                "divergence(c * gradient(c**2))", "divergence(vector_laplace(gradient(c))) * (0.1 - 0.2*I)"):
        eq = pde.PDE({"c": rhs}, bc=bc)
        ref = eq.solve(field, t_range=0.01, dt=1e-3, backend="torch", tracker=None)
        res = eq.solve(field, t_range=0.01, dt=1e-3, backend="hip", tracker=None)
        assert max_rel(np.array(res.data), ref.data) < 1e-10, rhs

    class Restated(pde.PDEBase):
        complex_valued = True

        def evolution_rate(self, state, t=0):
            grad = state.gradient(bc)
            div = ((1 + 0.5j) * grad).divergence(bc)
            dot = np.einsum("i...,i...->...", grad.data, grad.data.conjugate())
            return pde.ScalarField(state.grid, 1j * div.data + 0.1 * dot - 0.1 * state.data)

    for solver, adaptive in (("runge-kutta", False), ("runge-kutta", True), ("euler", True)):
        kw = dict(t_range=0.02, dt=1e-3, solver=solver, adaptive=adaptive, tracker=None, ret_info=True)
        ref, iref = Restated().solve(field, backend="numpy", **kw)
        res, info = pde.PDE({"c": first}, bc=bc).solve(field, backend="hip", **kw)
        assert info["solver"]["steps"] == iref["solver"]["steps"]
        assert max_rel(np.array(res.data), ref.data) < 1e-10, (solver, adaptive)


def test_adaptive_loops_of_complex_states_run_in_c(hip, monkeypatch):
    """Round 5 (VERDICT r4 "missing #4"): RKF45 and the adaptive Euler loop of a complex state inside ONE C call (`pdehip_jit_rk_run` /
    `pdehip_jit_euler_adaptive_run`, stage_fuse bit 1: modulus norm from an explicit error field) - the same bits and the same
    attempts as the loop driven from Python (PDEHIP_EXPR_LOOP=0), which is the one pinned against the reference above."""
    from pde_hip import expr as expr_mod

    calls = []
    original = expr_mod.SystemRhs.rk_run

    def recording(self, *args, **kwargs):
        res = original(self, *args, **kwargs)
        calls.append((kwargs.get("euler_adaptive", False), res is not None, len(args[2])))
        return res

    monkeypatch.setattr(expr_mod.SystemRhs, "rk_run", recording)
    grid = pde.UnitGrid([10, 8], periodic=[True, False])
    rng = np.random.default_rng(8)
    field = pde.ScalarField(grid, rng.uniform(-0.5, 0.5, grid.shape) + 1j * rng.uniform(-0.5, 0.5, grid.shape))
    eq = pde.PDE({"c": "-I * laplace(c) + (0.3 - 0.8*I) * c * Abs(c)**2"}, bc={"x": "periodic", "y": {"derivative": 0.1 - 0.2j}})
    for solver, nwork in (("runge-kutta", 8), ("euler", 4)):
        kw = dict(t_range=0.2, dt=1e-3, solver=solver, adaptive=True, tracker=None, ret_info=True)
        calls.clear()
        res_c, info_c = eq.solve(field, backend="hip", **kw)
        assert calls and all(c == (solver == "euler", True, nwork) for c in calls), calls
        monkeypatch.setenv("PDEHIP_EXPR_LOOP", "0")
        calls.clear()
        res_py, info_py = eq.solve(field, backend="hip", **kw)
        monkeypatch.delenv("PDEHIP_EXPR_LOOP")
        assert not calls
        assert info_c["solver"]["steps"] == info_py["solver"]["steps"] > 3
        np.testing.assert_array_equal(np.array(res_c.data), np.array(res_py.data))
        assert info_c["solver"]["dt_statistics"]["count"] == info_py["solver"]["dt_statistics"]["count"]


_ROBIN = {"x-": {"type": "mixed", "value": 0.5 + 1.5j, "const": 0.2 - 0.3j}, "x+": {"value": 1 + 2j}, "y": "periodic"}


@pytest.mark.parametrize("solver,adaptive", [("euler", False), ("runge-kutta", False), ("runge-kutta", True), ("euler", True)])
@pytest.mark.parametrize("cid", ["schroedinger_robin_2d", "schroedinger_robin_3d", "expression_complex_slope_2d"])
def test_mixed_conditions_with_complex_coefficients(hip, cid, solver, adaptive):
    """VERDICT r5 "missing" #3: a mixed (Robin) condition with a COMPLEX coefficient of the field value has a complex factor in its virtual
    point (pde/grids/boundaries/local.py:1927-1938): the ghost cells of either part depend on both parts.  The coupling terms are differences
    of two more applications of the stencil to the other part (pde_hip/complex_expr.py: COUPLING_SUFFIXES).  Against the reference's own run
    (tests/golden/complex.npz, made by make_golden_complex.py): equal step counts, <= 1e-10.  `expression_complex_slope_2d`: the same for a
    condition given as an EXPRESSION of time whose slope with respect to `value` is complex (refreshed inside the C loops)."""
    import json

    gold = np.load(Path(__file__).resolve().parent / "golden" / "complex.npz")
    case = {c["id"]: c for c in json.loads(str(gold["cases"]))}[cid]

    def cplx(v):
        return complex(v[0], v[1]) if isinstance(v, list) else v

    bc = {k: ({kk: cplx(vv) for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in case["bc"].items()}
    grid = pde.UnitGrid(case["shape"], periodic=case["periodic"])
    res, info = pde.PDE({case["var"]: case["rhs"]}, bc=bc).solve(pde.ScalarField(grid, gold[f"{cid}/input"]), t_range=case["t_range"], dt=case["dt"], solver=solver,
                                                                adaptive=adaptive, backend="hip", tracker=None, ret_info=True)
    key = f"{cid}/{solver}{'_adaptive' if adaptive else ''}"
    assert info["solver"]["steps"] == int(gold[f"{key}/steps"])
    assert max_rel(np.array(res.data), gold[f"{key}/final"]) < 1e-10


def test_operators_with_complex_factor_conditions(hip):
    """`field.laplace`, `gradient`, `make_operator` on complex data with a complex Robin coefficient: the coupling terms on the host-array path."""
    grid = pde.CartesianGrid([[0, 4], [0, 4.5]], [8, 9], periodic=[False, True])
    rng = np.random.default_rng(9)
    field = pde.ScalarField(grid, rng.uniform(-1, 1, grid.shape) + 1j * rng.uniform(-1, 1, grid.shape))
    np.testing.assert_allclose(field.laplace(_ROBIN, backend="hip").data, field.laplace(_ROBIN, backend="scipy").data, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(field.gradient(_ROBIN, backend="hip").data, field.gradient(_ROBIN, backend="scipy").data, rtol=1e-12, atol=1e-12)
    op = grid.make_operator("laplace", bc=_ROBIN, backend="hip", dtype=complex)
    np.testing.assert_allclose(op(field.data), field.laplace(_ROBIN, backend="scipy").data, rtol=1e-12, atol=1e-12)
    # a nonlinear equation on top: the reference's numpy backend against the expression kernels
    class Eq(pde.PDEBase):
        complex_valued = True

        def evolution_rate(self, state, t=0):
            c = state.data
            return pde.ScalarField(state.grid, 1j * state.laplace(_ROBIN).data - 0.1 * c * np.abs(c) ** 2 + 0.05 * state.laplace(_ROBIN).data)

    ref = Eq().solve(field, t_range=0.02, dt=1e-3, solver="runge-kutta", backend="numpy", tracker=None)
    res = pde.PDE({"c": "I * laplace(c) - 0.1 * c * Abs(c)**2 + 0.05 * laplace(c)"}, bc=_ROBIN).solve(field, t_range=0.02, dt=1e-3, solver="runge-kutta", backend="hip", tracker=None)
    assert max_rel(np.array(res.data), ref.data) < 1e-10


def test_what_is_refused(hip):
    grid = pde.UnitGrid([6, 6])
    field = pde.ScalarField(grid, 1.0 + 1j)
    with pytest.raises((NotImplementedError, RuntimeError)):   # gradient_squared is not linear: no coupling terms for complex-factor conditions
        pde.PDE({"c": "I * gradient_squared(c)"}, bc={"type": "mixed", "value": 1j, "const": 1}).solve(field, t_range=0.01, dt=1e-3, backend="hip", tracker=None)
    with pytest.raises((NotImplementedError, RuntimeError)):   # an expression condition of a complex field that is not affine in the field value
        pde.PDE({"c": "I * laplace(c)"}, bc={"x-": {"value_expression": "value**2 + I"}, "x+": {"value": 0}, "y": {"value": 0}}).solve(
            field, t_range=0.01, dt=1e-3, backend="hip", tracker=None)
    with pytest.raises((NotImplementedError, RuntimeError), match="must be real"):   # a complex array constant (its imaginary part was dropped)
        pde.PDE({"c": "I * laplace(c) + w * c"}, consts={"w": np.full(grid.shape, 1 + 2j)}).solve(field, t_range=0.01, dt=1e-3, backend="hip", tracker=None)
    with pytest.raises((NotImplementedError, RuntimeError)):   # tensors built from complex vectors inside an expression
        pde.PDE({"c": "dot(gradient(c), dot(vector_gradient(gradient(c)), gradient(c))) + I * c"}).solve(field, t_range=0.01, dt=1e-3, backend="hip", tracker=None)
