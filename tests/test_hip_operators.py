"""GPU parity tests: libpdehip (through the C ABI / host mirror) vs the CPU oracle and the goldens.

Bar: fp64 results are BIT-EXACT against the oracle (same expression order, -ffp-contract=off on
both sides) and therefore bit-exact against the reference's eager torch-CPU backend; vs the
reference scipy backend rtol 1e-12.  fp32: bit-exact against the oracle (fp32 storage / fp64
registers on both sides), 2e-6 relative against the reference's pure-fp32 torch result.
"""

from __future__ import annotations

import ctypes as C
import json

import numpy as np
import pytest
from helpers import case_ids, face_mask, get_case, host_faces, interior, make_grid, max_rel, oracle_grid, to_full

import pde_hip
from oracle import pde_oracle as O
from pde_hip import _abi

pytestmark = pytest.mark.gpu

OPS = case_ids("ops.npz")


@pytest.fixture(scope="module")
def backend():
    return pde_hip.get_backend("hip")


def _dev(backend, grid, dtype, comp_shape=()):
    from pde_hip.device import DeviceArray

    return DeviceArray(backend.grid_info(grid, dtype), comp_shape)


def test_library_loaded(backend):
    """The native library is the thing that runs (no fallback)."""
    assert backend._lib.path.name == "libpdehip.so"
    assert "gfx950" in backend.device_name or "MI3" in backend.device_name, backend.device_name


@pytest.mark.parametrize("cid", OPS)
def test_ghost_cells_vs_reference(backend, golden_ops, cid):
    case = get_case(golden_ops, cid)
    grid = make_grid(case)
    dtype = np.dtype(case.get("dtype", "float64"))
    dev = _dev(backend, grid, dtype).set_valid(golden_ops[f"{cid}/input"])
    bcs = grid.get_boundary_conditions(case["bc"], rank=0)
    backend.make_ghost_cell_setter(bcs)(dev)
    full = dev.get_hostfull()
    mask = face_mask(grid)
    np.testing.assert_array_equal(full[mask], golden_ops[f"{cid}/full"][mask])
    # vector field
    vdev = _dev(backend, grid, dtype, (grid.dim,)).set_valid(golden_ops[f"{cid}/vector_input"])
    backend.make_ghost_cell_setter(grid.get_boundary_conditions("auto_periodic_neumann", rank=1))(vdev)
    vmask = face_mask(grid, (grid.dim,))
    np.testing.assert_array_equal(vdev.get_hostfull()[vmask], golden_ops[f"{cid}/vector_full"][vmask])


def test_normal_bc(backend, golden_ops):
    grid = pde_hip.UnitGrid([4, 5], periodic=[False, True])
    bcs = grid.get_boundary_conditions(json.loads(str(golden_ops["normal_bc/bc"])), rank=1)
    dev = _dev(backend, grid, np.float64, (2,)).set_valid(golden_ops["normal_bc/input"])
    backend.make_ghost_cell_setter(bcs)(dev)
    mask = face_mask(grid, (2,)).copy()
    mask[1, 0, :] = mask[1, -1, :] = False
    np.testing.assert_array_equal(dev.get_hostfull()[mask], golden_ops["normal_bc/full"][mask])


@pytest.mark.parametrize("cid", OPS)
def test_operators_vs_golden_and_oracle(backend, golden_ops, cid):
    """ScalarField.laplace()/gradient()/... through the mirror API == reference torch-CPU results."""
    case = get_case(golden_ops, cid)
    grid = make_grid(case)
    dtype = np.dtype(case.get("dtype", "float64"))
    field = pde_hip.ScalarField(grid, golden_ops[f"{cid}/input"], dtype=dtype)
    g = oracle_grid(grid, dtype)
    full = golden_ops[f"{cid}/full"].copy()

    def check(out, oracle_out, ref_key, exact_ref=True):
        np.testing.assert_array_equal(out, oracle_out)  # HIP == oracle, bit for bit
        if ref_key in golden_ops:
            ref = golden_ops[ref_key]
            if dtype == np.float32:
                assert max_rel(out.astype(np.float64), ref.astype(np.float64)) < 2e-6
            elif exact_ref:
                np.testing.assert_array_equal(out, ref)
            else:
                np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(ref).max()))

    check(field.laplace(case["bc"]).data, O.laplace(g, full), f"{cid}/laplace_torch")
    check(field.laplace(case["bc"]).data, O.laplace(g, full), f"{cid}/laplace_scipy", exact_ref=False)
    check(field.gradient(case["bc"]).data, O.gradient(g, full), f"{cid}/gradient_central_torch", exact_ref=grid.dim > 1)
    for method in ["forward", "backward"]:
        check(field.gradient(case["bc"], method=method).data, O.gradient(g, full, method), f"{cid}/gradient_{method}_scipy", exact_ref=False)
    check(field.gradient_squared(case["bc"]).data, O.gradient_squared(g, full, True), f"{cid}/gradient_squared_central_torch", exact_ref=False)
    check(field.gradient_squared(case["bc"], central=False).data, O.gradient_squared(g, full, False), f"{cid}/gradient_squared_noncentral_torch", exact_ref=False)

    vec = pde_hip.VectorField(grid, golden_ops[f"{cid}/vector_input"], dtype=dtype)
    vfull = golden_ops[f"{cid}/vector_full"].copy()
    bc = "auto_periodic_neumann"
    check(vec.divergence(bc).data, O.divergence(g, vfull), f"{cid}/divergence_central_torch", exact_ref=False)
    for method in ["forward", "backward"]:
        check(vec.divergence(bc, method=method).data, O.divergence(g, vfull, method), f"{cid}/divergence_{method}_scipy", exact_ref=False)
    vlap = np.stack([O.laplace(g, np.ascontiguousarray(vfull[i])) for i in range(grid.dim)])
    check(vec.laplace(bc).data, vlap, f"{cid}/vector_laplace_torch")
    vgrad = np.stack([O.gradient(g, np.ascontiguousarray(vfull[i])) for i in range(grid.dim)])
    check(vec.gradient(bc).data, vgrad, f"{cid}/vector_gradient_torch", exact_ref=grid.dim > 1)


def test_apply_operator_no_bc_path(backend, golden_ops):
    """`_apply_operator` on reference-layout full arrays with a strided `out` view (fields/datafield_base.py:958-961)."""
    cid = "3d_mixed_bcs"
    case = get_case(golden_ops, cid)
    grid = make_grid(case)
    op = backend.make_operator_no_bc(grid, "laplace")
    out_field = pde_hip.ScalarField(grid, "empty")
    backend._apply_operator(op, golden_ops[f"{cid}/full"], out=out_field.data, grid=grid)
    assert not out_field.data.flags.c_contiguous
    np.testing.assert_array_equal(out_field.data, golden_ops[f"{cid}/laplace_torch"])


def test_error_behaviour(backend):
    grid = pde_hip.UnitGrid([4, 4])
    op = grid.make_operator("laplace", "auto_periodic_neumann")
    with pytest.raises(ValueError, match="Incompatible shapes"):
        op(np.zeros((3, 4)))
    with pytest.raises(ValueError, match="Incompatible shapes"):
        op(np.zeros((4, 4)), out=np.zeros((4, 5)))
    with pytest.raises(NotImplementedError, match="does not define operator"):
        grid.make_operator("no_such_operator", "auto_periodic_neumann")
    with pytest.raises(ValueError, match="Unknown derivative type"):
        grid.make_operator("gradient", "auto_periodic_neumann", method="sideways")
    with pytest.raises(NotImplementedError):
        backend.grid_info(grid, np.complex128)
    # C-ABI level errors carry the message of pdehip_last_error()
    lib = backend._lib
    g = _abi.make_grid((4, 4), (1.0, 1.0), np.float64)
    with pytest.raises(ValueError, match="NULL"):
        lib.laplace(C.byref(g), None, None, _abi.OUT_FULL, None)
    g.ndim = 7
    with pytest.raises(NotImplementedError, match="unsupported number of axes"):
        lib.laplace(C.byref(g), None, None, _abi.OUT_FULL, None)


SHAPES = [
    ((64, 64), np.float64), ((1024, 1024), np.float64), ((130, 258), np.float64), ((33, 77), np.float64),
    ((32, 32, 128), np.float64), ((48, 40, 256), np.float64), ((17, 9, 130), np.float64), ((8, 8, 12), np.float64), ((5, 7, 9), np.float64),
    ((64, 64, 64), np.float32), ((16, 24, 256), np.float32), ((256, 512), np.float32), ((4097,), np.float64), ((1000,), np.float32),
    ((9, 12, 150), np.float64), ((6, 10, 280), np.float32),      # split rows: 22 / 24 columns beyond whole chunks as a strip of the same launch
]


@pytest.mark.parametrize("shape,dtype", SHAPES)
@pytest.mark.parametrize("periodic", [True, False])
def test_laplace_family_vs_oracle_seeded(backend, rng, shape, dtype, periodic):
    """All fused laplace epilogues (plain / scaled / Euler / CH-mu), fast + generic kernels, bit-exact."""
    from pde_hip.device import DeviceArray

    dx = [0.5 + 0.25 * a for a in range(len(shape))]
    grid = pde_hip.CartesianGrid([[0, n * d] for n, d in zip(shape, dx)], shape, periodic=periodic)
    bc = "auto_periodic_neumann" if periodic else {"value": 0.3}
    bcs = grid.get_boundary_conditions(bc)
    data = rng.uniform(-1, 1, shape).astype(dtype)
    ydata = rng.uniform(-1, 1, shape).astype(dtype)
    g = oracle_grid(grid, dtype)
    full = to_full(grid, data)
    O.set_ghost_cells(g, 1, host_faces(bcs).c, full)
    yfull = to_full(grid, ydata)

    info = backend.grid_info(grid, dtype)
    lib = backend._lib
    dev = DeviceArray(info).set_valid(data)
    ydev = DeviceArray(info).set_valid(ydata)
    out = DeviceArray(info)
    backend.make_ghost_cell_setter(bcs)(dev)

    lib.laplace(info.ref, dev.ptr, out.ptr, _abi.OUT_FULL, None)
    np.testing.assert_array_equal(out.get_valid(), O.laplace(g, full))
    lib.laplace_scaled(info.ref, dev.ptr, out.ptr, 0.7, 0.01, None)
    np.testing.assert_array_equal(out.get_valid(), interior(grid, O.laplace_scaled(g, full, 0.7, 0.01)))
    lib.laplace_euler(info.ref, dev.ptr, dev.ptr, out.ptr, 0.7, 0.01, None)
    np.testing.assert_array_equal(out.get_valid(), interior(grid, O.laplace_euler(g, full, full, 0.7, 0.01)))
    lib.laplace_euler(info.ref, dev.ptr, ydev.ptr, out.ptr, 1.0, 0.02, None)
    np.testing.assert_array_equal(out.get_valid(), interior(grid, O.laplace_euler(g, full, yfull, 1.0, 0.02)))
    lib.cahn_hilliard_mu(info.ref, dev.ptr, out.ptr, 0.9, None)
    np.testing.assert_array_equal(out.get_valid(), interior(grid, O.cahn_hilliard_mu(g, full, 0.9)))
    # valid-layout output through the C ABI (the reference's (arr_full, out_valid) signature)
    from pde_hip.device import DeviceBuffer

    vbuf = DeviceBuffer(data.nbytes)
    lib.laplace(info.ref, dev.ptr, vbuf.ptr, _abi.OUT_VALID, None)
    host = np.empty(shape, dtype)
    lib.memcpy_d2h(host.ctypes.data, vbuf.ptr, host.nbytes, None)
    np.testing.assert_array_equal(host, O.laplace(g, full))


def test_fast_and_generic_kernels_agree(backend, rng, monkeypatch):
    """The register-pipelined kernel and the one-cell-per-thread fallback give identical bits."""
    import subprocess
    import sys

    code = (
        "import sys, numpy as np; sys.path[:0]=['py-pde_amd','tests','.'];"
        "import pde_hip; from pde_hip.device import DeviceArray; from pde_hip import _abi;"
        "b=pde_hip.get_backend('hip'); g=pde_hip.UnitGrid([24,40,128],periodic=True);"
        "d=np.random.default_rng(3).uniform(-1,1,g.shape); i=b.grid_info(g,np.float64);"
        "a=DeviceArray(i).set_valid(d); o=DeviceArray(i);"
        "b.make_ghost_cell_setter(g.get_boundary_conditions('periodic'))(a);"
        "b._lib.laplace(i.ref,a.ptr,o.ptr,_abi.OUT_FULL,None); np.save(sys.argv[1], o.get_valid())"
    )
    import os
    import tempfile

    outs = []
    for force in ("0", "1"):
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "o.npy")
            env = dict(os.environ, PDEHIP_FORCE_GENERIC=force)
            subprocess.run([sys.executable, "-c", code, path], check=True, env=env, cwd=str(__import__("helpers").ROOT))
            outs.append(np.load(path))
    np.testing.assert_array_equal(outs[0], outs[1])


def test_known_answers_on_device(backend):
    """Reference known answers (tests/fields/test_scalar_fields.py:108-139, numba operator tests :118-170)."""
    grid = pde_hip.CartesianGrid([[0, 2 * np.pi]] * 2, 16, periodic=True)
    s = pde_hip.ScalarField.from_expression(grid, "sin(x) + cos(y)")
    lap = s.laplace("auto_periodic_neumann")
    np.testing.assert_allclose(lap.data, -s.data, rtol=0.1, atol=0.1)
    s.laplace("auto_periodic_neumann", out=lap)  # `out=` reuse path
    np.testing.assert_allclose(lap.data, -s.data, rtol=0.1, atol=0.1)
    const = pde_hip.ScalarField(grid, 3.0)
    np.testing.assert_allclose(const.laplace("periodic").data, 0, atol=1e-10)
    np.testing.assert_allclose(const.gradient("periodic").data, 0, atol=1e-10)
    # div(grad) == laplace up to the wider stencil: linear field -> 0 for both
    g1 = pde_hip.CartesianGrid([[0, 1]], 32)
    lin = pde_hip.ScalarField.from_expression(g1, "2*x")
    np.testing.assert_allclose(lin.laplace({"x-": {"derivative": -2.0}, "x+": {"derivative": 2.0}}).data, 0, atol=1e-10)
    sq = pde_hip.ScalarField.from_expression(g1, "x**2")
    np.testing.assert_allclose(sq.laplace({"curvature": 2.0}).data, 2.0, rtol=1e-9)


def test_full_size_properties_512cubed(backend):
    """BASELINE size (512^3 fp64): size-independent properties instead of a full oracle run.

    * a slab of the result equals the oracle evaluated on that slab alone (stencil locality),
    * linearity: lap(a*u + v) == a*lap(u) + lap(v) to rounding,
    * the periodic Laplacian of any field sums to ~0 (discrete divergence theorem).
    """
    from pde_hip.device import DeviceArray

    n = 512
    grid = pde_hip.UnitGrid([n, n, n], periodic=True)
    rng = np.random.default_rng(0)
    u = rng.random((n, n, n))
    info = backend.grid_info(grid, np.float64)
    bcs = grid.get_boundary_conditions("auto_periodic_neumann")
    setter = backend.make_ghost_cell_setter(bcs)
    du = DeviceArray(info).set_valid(u)
    out = DeviceArray(info)
    setter(du)
    backend._lib.laplace(info.ref, du.ptr, out.ptr, _abi.OUT_FULL, None)
    lap_u = out.get_valid()
    # (1) oracle on slabs (with their neighbouring layers as ghost layers, periodic in y/z)
    for lo in (0, 200, n - 6):
        sl = slice(lo, lo + 6)
        sub = pde_hip.CartesianGrid([[0, 6], [0, n], [0, n]], [6, n, n], periodic=[False, True, True])
        g = oracle_grid(sub)
        full = to_full(sub, u[sl])
        full[0, 1:-1, 1:-1] = u[(lo - 1) % n]
        full[-1, 1:-1, 1:-1] = u[(lo + 6) % n]
        faces = host_faces(sub.get_boundary_conditions({"x": {"value": 0}, "y": "periodic", "z": "periodic"}), skip={(0, False), (0, True)})
        O.set_ghost_cells(g, 1, faces.c, full)
        np.testing.assert_array_equal(lap_u[sl], O.laplace(g, full))
    # (2) sum rule
    assert abs(lap_u.sum()) < 1e-6 * np.abs(lap_u).sum()
    # (3) linearity
    v = rng.random((n, n, n))
    dv = DeviceArray(info).set_valid(2.5 * u + v)
    setter(dv)
    backend._lib.laplace(info.ref, dv.ptr, out.ptr, _abi.OUT_FULL, None)
    lap_c = out.get_valid()
    dv.set_valid(v)
    setter(dv)
    backend._lib.laplace(info.ref, dv.ptr, out.ptr, _abi.OUT_FULL, None)
    assert max_rel(lap_c, 2.5 * lap_u + out.get_valid()) < 1e-13


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape,bounds", [((64,), [[0, 6.0]]), ((48, 40), [[0, 4.0], [0, 3.0]]), ((33, 27), [[0, 3.3], [0, 2.7]]), ((512, 512), [[0, 512.0]] * 2)])
def test_spectral_laplace_against_numpy_fft(rng, shape, bounds, dtype):
    """`spectral=True` (pde/backends/numba/operators/cartesian.py:232-330): hipFFT + the factor table against the reference's formula
    evaluated with numpy's FFT in double, and - small grids - against the oracle's plain DFT."""
    backend = pde_hip.get_backend("hip")
    grid = pde_hip.CartesianGrid(bounds, list(shape), periodic=True)
    data = rng.uniform(-1, 1, shape).astype(dtype)
    got = pde_hip.ScalarField(grid, data, dtype=dtype).laplace("periodic", backend=backend, spectral=True).data
    ks = [np.fft.fftfreq(n, d) for n, d in zip(grid.shape, grid.discretization)]
    d64 = data.astype(np.float64)
    if len(shape) == 1:
        expect = np.fft.ifft(-((2 * np.pi * ks[0]) ** 2) * np.fft.fft(d64)).real
    else:
        expect = np.fft.ifft2(-4 * np.pi**2 * (ks[0][:, None] ** 2 + ks[1][None, :] ** 2) * np.fft.fft2(d64)).real
    tol = 1e-12 if dtype == np.float64 else 2e-5
    assert got.dtype == dtype and max_rel(got, expect) < tol
    if np.prod(shape) <= 4096:
        assert max_rel(O.laplace_spectral(oracle_grid(grid, dtype), to_full(grid, data)), expect) < tol
    with pytest.raises(NotImplementedError):
        pde_hip.ScalarField(pde_hip.UnitGrid([8, 8, 8], periodic=True), 1.0).laplace("periodic", backend=backend, spectral=True)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(7, 9), (5, 6, 130)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_field_products_and_expression_functions(shape, dtype):
    """`pdehip_field_product` (make_inner_prod_operator / make_outer_prod_operator, pde/backends/base.py:567-610) against numpy's einsum for
    every rank combination, real and complex, conjugated or not; `make_expression_function` (:653-676) - pointwise expressions of arrays
    and numbers through the run-time compiled kernels - against numpy."""
    import sympy as sp

    backend = pde_hip.get_backend("hip")
    grid = pde_hip.UnitGrid(list(shape), periodic=True)
    d = len(shape)
    rng = np.random.default_rng(12)
    tol = 1e-13 if dtype == np.float64 else 2e-6
    cdtype = np.complex128 if dtype == np.float64 else np.complex64
    v1, v2 = rng.random((d, *shape)).astype(dtype), rng.random((d, *shape)).astype(dtype)
    t1 = (rng.random((d, d, *shape)) + 1j * rng.random((d, d, *shape))).astype(cdtype)
    t2 = rng.random((d, d, *shape)).astype(dtype)
    for conj in (True, False):
        dot = backend._make_product(grid, False, conj)
        cj = (lambda x: x.conj()) if conj else (lambda x: x)
        np.testing.assert_allclose(dot(v1, v2), np.einsum("i...,i...->...", v1, v2), rtol=tol)
        np.testing.assert_allclose(dot(t2, v1), np.einsum("ij...,j...->i...", t2, v1), rtol=tol)
        np.testing.assert_allclose(dot(v1, t1), np.einsum("i...,ij...->j...", v1, cj(t1)), rtol=tol)
        np.testing.assert_allclose(dot(t1, t1), np.einsum("ij...,jk...->ik...", t1, cj(t1)), rtol=tol)
    np.testing.assert_allclose(backend._make_product(grid, True, False)(v1, v2), np.einsum("i...,j...->ij...", v1, v2), rtol=tol)

    class Expr:       # the attributes of pde.tools.expressions.ExpressionBase that the backend reads
        def __init__(self, text, names, consts=None, funcs=None):
            self._sympy_expr, self.vars, self.consts, self.user_funcs = sp.sympify(text), tuple(names), dict(consts or {}), dict(funcs or {})

    a, b = rng.random(shape).astype(dtype), rng.random(shape).astype(dtype)
    f = backend.make_expression_function(Expr("sin(a) * b + s * a**2 - sq(b) + k", ["a", "b", "s"], {"k": 0.25}, {"sq": lambda x: x * x}))
    for sval in (0.7, -1.5):
        np.testing.assert_allclose(f(a, b, sval), np.sin(a.astype(float)) * b + sval * a.astype(float) ** 2 - b.astype(float) ** 2 + 0.25, rtol=10 * tol)
    f1 = backend.make_expression_function(Expr("a * b + 2", ["a", "b"]), single_arg=True)
    np.testing.assert_allclose(f1(np.stack([a, b])), a.astype(float) * b + 2, rtol=10 * tol)
    # what `pde.tools.expressions.evaluate` asks for (pde/tools/expressions.py:986-1080): operators of this backend as user functions
    # (called as `op(arg, none, bc_args)`), results of any rank, products expanded symbolically, complex fields split into their parts
    bc = {"value": 0.3} if len(shape) == 1 else "auto_periodic_neumann"
    ops = {"laplace": backend.make_operator(grid, "laplace", bcs=grid.get_boundary_conditions(bc, rank=0)),
           "gradient": backend.make_operator(grid, "gradient", bcs=grid.get_boundary_conditions(bc, rank=0)),
           "dot": backend._make_product(grid, False, True), "outer": backend._make_product(grid, True, False)}
    sig = ["a", "b", "none", "bc_args"]
    lap = ops["laplace"](a)
    grad = ops["gradient"](a)
    call = lambda text, *arrays: backend.make_expression_function(Expr(text, sig), user_funcs=ops)(*arrays, None, {})    # noqa: E731
    np.testing.assert_allclose(call("laplace(a, none, bc_args) * b + 2 * a", a, b), lap.astype(float) * b + 2 * a.astype(float), rtol=20 * tol, atol=20 * tol)
    np.testing.assert_allclose(call("laplace(a**2 + b, none, bc_args)", a, b), ops["laplace"]((a.astype(float) ** 2 + b).astype(dtype)), rtol=50 * tol, atol=50 * tol)
    res = call("b * gradient(a, none, bc_args)", a, b)
    assert res.shape == (len(shape), *shape)
    np.testing.assert_allclose(res, b.astype(float) * grad, rtol=20 * tol, atol=20 * tol)
    np.testing.assert_allclose(call("dot(gradient(a, none, bc_args), gradient(a, none, bc_args))", a, b), np.einsum("i...,i...->...", grad, grad).astype(float), rtol=50 * tol, atol=50 * tol)
    np.testing.assert_allclose(call("outer(gradient(a, none, bc_args), gradient(b, none, bc_args))", a, b),
                               np.einsum("i...,j...->ij...", grad, ops["gradient"](b)).astype(float), rtol=50 * tol, atol=50 * tol)
    z = (a + 1j * b).astype(np.complex128 if dtype == np.float64 else np.complex64)
    np.testing.assert_allclose(call("abs(a)**2 + b", z, b), np.abs(z.astype(complex)) ** 2 + b, rtol=20 * tol)
    got = call("(1 + 2*I) * a * b", z, b)
    assert np.iscomplexobj(got)
    np.testing.assert_allclose(got, (1 + 2j) * z.astype(complex) * b, rtol=20 * tol)
