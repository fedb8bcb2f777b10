"""CPU tests of the host-side logic above the C ABI (no GPU, no compute calls)."""

from __future__ import annotations

import math

import numpy as np
import pytest
from helpers import host_faces

import pde_hip
from pde_hip import _abi
from pde_hip.backend import _match_expression_rhs
from pde_hip.boundaries import BCDataError, BoundariesList
from pde_hip.solvers import OnlineStatistics, make_dt_adjuster


def test_grid_geometry():
    g = pde_hip.CartesianGrid([[0, 3], [1, 4.5]], [6, 5], periodic=[False, True])
    assert g.dim == g.num_axes == 2 and g.shape == (6, 5) and g._shape_full == (8, 7)
    np.testing.assert_allclose(g.discretization, [0.5, 0.7])
    np.testing.assert_allclose(g.axes_coords[0], 0.25 + 0.5 * np.arange(6))
    assert g.periodic == [False, True] and g.axes == ["x", "y"]
    u = pde_hip.UnitGrid([4, 3, 2], periodic=True)
    assert tuple(u.discretization) == (1.0, 1.0, 1.0) and u.axes_bounds == ((0, 4), (0, 3), (0, 2))
    assert pde_hip.CartesianGrid([[0, 1]] * 2, 8).shape == (8, 8)
    with pytest.raises(ValueError):
        pde_hip.CartesianGrid([[0, 1]], [4, 4])


def test_field_memory_contract():
    """`.data` is a strided interior view of a C-contiguous full array (fields/base.py:116-160)."""
    g = pde_hip.UnitGrid([4, 5])
    f = pde_hip.ScalarField(g, np.arange(20.0).reshape(4, 5))
    assert f._data_full.shape == (6, 7) and f._data_full.flags.c_contiguous
    assert f.data.base is f._data_full and not f.data.flags.c_contiguous
    f.data[1, 2] = -1
    assert f._data_full[2, 3] == -1
    v = pde_hip.VectorField.random_uniform(g, rng=np.random.default_rng(0))
    assert v.data.shape == (2, 4, 5) and v._data_full.shape == (2, 6, 7)
    r = pde_hip.ScalarField.random_uniform(g, rng=np.random.default_rng(0))
    np.testing.assert_array_equal(r.data, np.random.default_rng(0).uniform(0, 1, (4, 5)))


def test_bc_virtual_point_data():
    """const / factor / index of each BC type (local.py:1728-1731, :1749-1753, :1773-1778, :1927-1938, :2081-2103)."""
    g = pde_hip.CartesianGrid([[0, 2]], [4], periodic=False)  # dx = 0.5
    mk = lambda data: g.get_boundary_conditions({"x-": data, "x+": data})[0]  # noqa: E731
    low, high = mk({"value": 3.0})
    assert low.get_virtual_point_data()[0] == 6.0 and float(low.get_virtual_point_data()[1]) == -1 and low.get_virtual_point_data()[2] == 0
    assert high.get_virtual_point_data()[2] == 3
    c, f, i = mk({"derivative": 2.0}).high.get_virtual_point_data()
    assert (float(c), float(f), i) == (1.0, 1.0, 3)
    c, f, i = mk({"type": "mixed", "value": 2.0, "const": 0.7}).low.get_virtual_point_data()
    np.testing.assert_allclose([c, f], [2 * 0.5 * 0.7 / 3, 1 / 3])
    c, f1, i1, f2, i2 = mk({"curvature": 0.4}).high.get_virtual_point_data()
    assert (float(c), float(f1), i1, float(f2), i2) == (0.4 * 0.25, 2.0, 3, -1.0, 2)
    gp = pde_hip.UnitGrid([4], periodic=True)
    lo, hi = gp.get_boundary_conditions("periodic")[0]
    assert lo.get_virtual_point_data() == (0.0, 1, 3) and hi.get_virtual_point_data() == (0.0, 1, 0)
    lo, _ = gp.get_boundary_conditions("anti-periodic")[0]
    assert lo.get_virtual_point_data() == (0.0, -1, 3)
    with pytest.raises(RuntimeError, match="at least 2 support points"):
        pde_hip.UnitGrid([1]).get_boundary_conditions("extrapolate")[0].low.get_virtual_point_data()


def test_bc_parsing_formats_and_errors():
    g = pde_hip.UnitGrid([4, 4], periodic=[True, False])
    bcs = g.get_boundary_conditions("auto_periodic_neumann")
    assert bcs.periodic == [True, False] and type(bcs[1].low).__name__ == "NeumannBC"
    assert BoundariesList.from_data(bcs, grid=g) is bcs
    b2 = g.get_boundary_conditions({"x": "periodic", "y-": {"value": 1}, "y+": "derivative"})
    assert type(b2[1].low).__name__ == "DirichletBC" and type(b2[1].high).__name__ == "NeumannBC"
    b3 = g.get_boundary_conditions(["periodic", [{"value": 1}, {"derivative": 2}]])
    assert float(b3[1].high.value) == 2.0
    b4 = g.get_boundary_conditions({"*": {"value": 0}, "x": "periodic"})
    assert type(b4[1].high).__name__ == "DirichletBC"
    with pytest.raises(BCDataError):
        g.get_boundary_conditions({"x": "periodic", "y": "no_such_condition"})
    with pytest.raises(RuntimeError, match="Periodicity"):
        g.get_boundary_conditions("periodic")  # y axis is not periodic
    with pytest.raises(BCDataError):
        g.get_boundary_conditions({"x": "periodic"})  # y missing
    with pytest.raises(ValueError):
        g.get_boundary_conditions({"x": "periodic", "y": {"value": [1, 2, 3]}})  # wrong face shape
    with pytest.raises(NotImplementedError):
        g.get_boundary_conditions(lambda data, args=None: None)


def test_face_table_conversion():
    """BoundariesList -> pdehip_bc_face_t[6] (scalars for homogeneous scalar BCs, arrays otherwise)."""
    g = pde_hip.CartesianGrid([[0, 2], [0, 3]], [4, 6], periodic=[False, True])
    bcs = g.get_boundary_conditions({"x-": {"value": 1.5}, "x+": {"derivative": np.arange(6.0)}, "y": "periodic"})
    t = host_faces(bcs)
    lo, hi = t.c[0], t.c[1]
    assert (lo.kind, lo.flags, lo.index1, lo.const_v, lo.factor1) == (_abi.BC_ORDER1, 0, 0, 3.0, -1.0)
    assert hi.flags == _abi.BCF_ARRAYS and hi.index1 == 3 and len(t.keepalive) == 2
    np.testing.assert_allclose(t.keepalive[0].arr, 0.5 * np.arange(6.0))
    assert (t.c[2].index1, t.c[3].index1, t.c[2].factor1) == (5, 0, 1.0)
    assert t.c[4].kind == _abi.BC_SKIP
    # faces replaced by a halo exchange are skipped
    t2 = host_faces(bcs, skip={(0, True)})
    assert t2.c[1].kind == _abi.BC_SKIP and t2.c[0].kind == _abi.BC_ORDER1
    # vector field with homogeneous per-component values -> arrays of shape (dim, face)
    vb = g.get_boundary_conditions({"x": {"value": [1.0, 2.0]}, "y": "periodic"}, rank=1)
    tv = host_faces(vb, (2,))
    assert tv.c[0].flags == _abi.BCF_ARRAYS and tv.keepalive[0].arr.shape == (2, 6)
    np.testing.assert_allclose(tv.keepalive[0].arr[:, 0], [2.0, 4.0])


def test_dt_adjuster_and_statistics():
    """solvers/base.py:559-592 and tools/math.py:125-174."""
    adj = make_dt_adjuster(1e-10, 1e10)
    assert adj(1.0, 1e-4) == 4.0
    assert adj(1.0, math.nan) == 0.25
    assert adj(1.0, 1.0) == pytest.approx(0.9)
    assert adj(1.0, 1e9) == pytest.approx(0.1)
    assert make_dt_adjuster(1e-10, 2.0)(1.0, 1e-4) == 2.0
    with pytest.raises(RuntimeError, match="Time step below"):
        make_dt_adjuster(0.5, 10)(1.0, 1e3)
    with pytest.raises(RuntimeError, match="Encountered NaN"):
        make_dt_adjuster(0.5, 10)(1.0, math.nan)
    s = OnlineStatistics()
    for v in [1.0, 2.0, 4.0]:
        s.add(v)
    d = s.to_dict()
    assert d["count"] == 3 and d["min"] == 1.0 and d["max"] == 4.0
    assert d["mean"] == pytest.approx(7 / 3) and d["std"] == pytest.approx(np.std([1, 2, 4], ddof=1))


def test_expression_matcher():
    assert _match_expression_rhs("laplace(c**3 - c - laplace(c))", "c", {}) == (_abi.RHS_CAHN_HILLIARD, 1.0)
    assert _match_expression_rhs("laplace(c**3 - c - 0.5*laplace(c))", "c", {}) == (_abi.RHS_CAHN_HILLIARD, 0.5)
    assert _match_expression_rhs("D * laplace(u)", "u", {"D": 2.0}) == (_abi.RHS_DIFFUSION, 2.0)
    assert _match_expression_rhs("laplace(c)", "c", {}) == (_abi.RHS_DIFFUSION, 1.0)
    assert _match_expression_rhs("∇²(c)", "c", {}) == (_abi.RHS_DIFFUSION, 1.0)
    assert _match_expression_rhs("laplace(c) + c", "c", {}) is None
    assert _match_expression_rhs("laplace(c**2)", "c", {}) is None


def test_operator_registry():
    """register_operator / get_operator_info walk backend x grid MROs (backends/base.py:256-376)."""
    from pde_hip.backend import HipBackend

    ops = HipBackend._operators[pde_hip.CartesianGrid]
    assert {"laplace", "gradient", "divergence", "gradient_squared", "vector_laplace", "vector_gradient", "tensor_divergence"} <= set(ops)
    assert (ops["gradient"].rank_in, ops["gradient"].rank_out) == (0, 1)
    assert (ops["divergence"].rank_in, ops["divergence"].rank_out) == (1, 0)
    dummy = object.__new__(HipBackend)
    dummy.name = "hip"
    info = dummy.get_operator_info(pde_hip.UnitGrid([4]), "laplace")  # UnitGrid inherits CartesianGrid's operators
    assert info.name == "laplace"
    with pytest.raises(NotImplementedError, match="does not define operator 'curl'"):
        dummy.get_operator_info(pde_hip.UnitGrid([4]), "curl")

    @HipBackend.register_operator(pde_hip.UnitGrid, "my_op", rank_in=0, rank_out=0)
    def make_my_op(grid, **kwargs):
        return lambda arr, out: None

    assert "my_op" in dummy.get_registered_operators(pde_hip.UnitGrid([4]))
    assert "my_op" not in dummy.get_registered_operators(pde_hip.CartesianGrid([[0, 1]], 4))
    del HipBackend._operators[pde_hip.UnitGrid]["my_op"]


def test_dtype_codes():
    assert _abi.dtype_code(np.float64) == _abi.F64 and _abi.dtype_code("float32") == _abi.F32
    with pytest.raises(NotImplementedError):
        _abi.dtype_code(np.complex128)
    with pytest.raises(NotImplementedError):
        _abi.make_grid((2, 2, 2, 2), (1, 1, 1, 1), np.float64)
