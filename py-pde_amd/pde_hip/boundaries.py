"""Constant boundary conditions reduced to virtual-point data.

Mirror of the part of ``pde.grids.boundaries`` the hot path consumes: every local condition is
``ghost = const + factor * arr[index]`` (1st order, ``pde/grids/boundaries/local.py:1611-1636``)
or ``const + f1 * arr[i1] + f2 * arr[i2]`` (2nd order, ``:2022-2061``).  The classes expose the
attribute names :class:`pde_hip.backend.HipBackend` reads from real py-pde objects as well
(``axis``, ``upper``, ``rank``, ``homogeneous``, ``normal``, ``get_virtual_point_data()``), so the
conversion to ``pdehip_bc_face_t`` is shared.  Expression conditions (``value_expression`` ...,
``pde/grids/boundaries/local.py:766-1150``) are :class:`ExpressionBC` here: they carry the sympy form of
their virtual point, which ``pde_hip/bc_expr.py`` lowers for the device.
"""

from __future__ import annotations

from typing import Any

import numpy as np


class BCDataError(ValueError):
    """Boundary data could not be interpreted (same name as ``pde.grids.boundaries.local``)."""


def _face_shape(grid, axis: int) -> tuple[int, ...]:
    return tuple(n for a, n in enumerate(grid.shape) if a != axis)


class BCBase:
    """One side of one axis."""

    names: list[str] = []
    normal = False
    _conditions: dict[str, type["BCBase"]] = {}

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        for name in cls.names:
            BCBase._conditions[name] = cls

    def __init__(self, grid, axis: int, upper: bool, *, rank: int = 0, value: Any = 0):
        self.grid = grid
        self.axis = int(axis)
        self.upper = bool(upper)
        self.rank = int(rank)
        if self.normal and self.rank < 1:
            msg = "Normal boundary conditions require a tensorial field"
            raise ValueError(msg)
        self.value = value  # type: ignore[assignment]

    # value handling ------------------------------------------------------------------------
    @property
    def _shape_tensor(self) -> tuple[int, ...]:
        rank = self.rank - 1 if self.normal else self.rank
        return (self.grid.dim,) * rank

    @property
    def value(self) -> np.ndarray:
        return self._value

    @value.setter
    def value(self, value) -> None:
        self._value, self.homogeneous = self._parse_value(value)

    def _parse_value(self, value) -> tuple[np.ndarray, bool]:
        """scalar / tensor → homogeneous; array incl. face shape or coordinate expression → not."""
        shape_t = self._shape_tensor
        shape_f = _face_shape(self.grid, self.axis)
        if isinstance(value, str):
            # expression of the coordinates along the boundary, evaluated once (local.py:1393-1431)
            import sympy

            names = [n for a, n in enumerate(self.grid.axes) if a != self.axis]
            coords = [c for a, c in enumerate(self.grid.axes_coords) if a != self.axis]
            expr = sympy.sympify(value)
            func = sympy.lambdify([sympy.Symbol(n) for n in names], expr, modules="numpy")
            mesh = np.meshgrid(*coords, indexing="ij") if coords else []
            arr = np.broadcast_to(np.asarray(func(*mesh), dtype=np.double), shape_t + shape_f)
            return np.array(arr), False
        arr = np.asarray(value)
        arr = arr.astype(np.complex128 if np.iscomplexobj(arr) and np.any(np.imag(arr) != 0) else np.double)   # (complex values: complex fields only)
        if arr.ndim <= len(shape_t):
            try:
                return np.array(np.broadcast_to(arr, shape_t)), True
            except ValueError:
                pass
        try:
            full = np.broadcast_to(arr, shape_t + shape_f)
        except ValueError:
            msg = f"Dimensions {arr.shape} of the given value are incompatible with the expected shape {shape_t + shape_f}"
            raise ValueError(msg) from None
        return np.array(full), False

    def get_virtual_point_data(self):
        raise NotImplementedError

    def __repr__(self) -> str:
        side = "upper" if self.upper else "lower"
        return f"{self.__class__.__name__}(axis={self.axis}, {side}, value={self.value!r})"

    @classmethod
    def from_data(cls, grid, axis: int, upper: bool, data, *, rank: int = 0) -> "BCBase":
        if isinstance(data, BCBase):
            return data
        if isinstance(data, str):
            if data not in cls._conditions:
                raise BCDataError(f"Boundary condition `{data}` not defined.")
            return cls._conditions[data](grid, axis, upper, rank=rank)
        if isinstance(data, dict):
            data = dict(data)
            if "type" in data:
                name = data.pop("type")
            else:
                keys = [k for k in data if k in cls._conditions]
                if len(keys) != 1:
                    raise BCDataError(f"Boundary conditions `{data}` could not be parsed.")
                name = keys[0]
                data["value"] = data.pop(name)
            if name not in cls._conditions:
                raise BCDataError(f"Boundary condition `{name}` not defined.")
            try:
                return cls._conditions[name](grid, axis, upper, rank=rank, **data)
            except TypeError as err:
                raise BCDataError(f"Unsupported boundary data `{data}`: {err}") from None
        raise BCDataError(f"Unsupported boundary format: `{data}`.")


class _PeriodicBC(BCBase):
    """One side of a periodic axis (local.py:1728-1731)."""

    def __init__(self, grid, axis, upper, *, rank=0, flip_sign=False):
        super().__init__(grid, axis, upper, rank=rank, value=0)
        self.flip_sign = bool(flip_sign)

    def get_virtual_point_data(self):
        index = 0 if self.upper else self.grid.shape[self.axis] - 1
        return (0.0, -1 if self.flip_sign else 1, index)


class DirichletBC(BCBase):
    names = ["value", "dirichlet"]

    def get_virtual_point_data(self):  # local.py:1749-1753
        const = 2 * self.value
        index = self.grid.shape[self.axis] - 1 if self.upper else 0
        return (const, -np.ones_like(const), index)


class NeumannBC(BCBase):
    names = ["derivative", "neumann"]

    def get_virtual_point_data(self):  # local.py:1773-1778
        dx = self.grid.discretization[self.axis]
        const = dx * self.value
        index = self.grid.shape[self.axis] - 1 if self.upper else 0
        return (const, np.ones_like(const), index)


class MixedBC(BCBase):
    names = ["mixed", "robin"]

    def __init__(self, grid, axis, upper, *, rank=0, value=0, const=0):
        super().__init__(grid, axis, upper, rank=rank, value=value)
        self.const, hom = self._parse_value(const)
        self.homogeneous = self.homogeneous and hom

    def get_virtual_point_data(self):  # local.py:1927-1938
        dx = self.grid.discretization[self.axis]
        with np.errstate(invalid="ignore", divide="ignore"):
            const = np.asarray(2 * dx * self.const / (2 + dx * self.value))
            factor = np.asarray((2 - dx * self.value) / (2 + dx * self.value))
        const, factor = np.broadcast_arrays(const, factor)
        const, factor = np.array(const), np.array(factor)
        const[~np.isfinite(factor)] = 0
        factor[~np.isfinite(factor)] = -1
        index = self.grid.shape[self.axis] - 1 if self.upper else 0
        return (const, factor, index)


class CurvatureBC(BCBase):
    names = ["curvature", "second_derivative", "extrapolate"]

    def get_virtual_point_data(self):  # local.py:2081-2103
        size = self.grid.shape[self.axis]
        dx = self.grid.discretization[self.axis]
        if size < 2:
            msg = "Need at least 2 support points to use curvature boundary condition"
            raise RuntimeError(msg)
        value = np.asarray(self.value * dx**2)
        f1 = np.full_like(value, 2.0)
        f2 = np.full_like(value, -1.0)
        i1, i2 = (size - 1, size - 2) if self.upper else (0, 1)
        return (value, f1, i1, f2, i2)


class ExpressionBC(BCBase):
    """Virtual point given by an expression ``F(value, dx, *coords, t)`` of the field value in the adjacent cell (or in
    ``value_cell``), the spacing normal to the wall, the wall point and the time (``pde/grids/boundaries/local.py:766-866``):
    ``virtual_point`` = the expression itself, ``value`` -> ``2*(e) - value``, ``derivative`` -> ``dx*(e) + value``,
    ``mixed`` -> ``(2*dx*(const) + (2 - (e)*dx)*value) / ((e)*dx + 2)``.  Strings only (Python functions cannot travel to
    the device through this mirror)."""

    names = ["virtual_point"]
    target = "virtual_point"

    def __init__(self, grid, axis, upper, *, rank=0, value=0, const=0, target=None, value_cell=None):
        BCBase.__init__(self, grid, axis, upper, rank=rank, value=0)
        if self.rank != 0:
            msg = "Expression boundary conditions only work for scalar conditions"
            raise NotImplementedError(msg)
        if callable(value) or callable(const):
            msg = "pde_hip mirror: expression boundary conditions take strings or numbers"
            raise NotImplementedError(msg)
        self.target = self.target if target is None else target
        self.value_cell = value_cell
        self.homogeneous = False
        self._input = {"value_expr": value, "const_expr": const, "target": self.target}
        if self.target == "virtual_point":
            text = f"{value}"
        elif self.target == "value":
            text = f"2 * ({value}) - value"
        elif self.target == "derivative":
            text = f"dx * ({value}) + value"
        elif self.target == "mixed":
            text = f"(2 * dx * ({const}) + (2 - ({value}) * dx) * value) / (({value}) * dx + 2)"
        else:
            msg = f"Unknown target `{self.target}` for expression"
            raise ValueError(msg)
        import sympy

        names = ["value", "dx", *grid.axes, "t"]
        try:
            self.virtual_point_sympy = sympy.sympify(text.replace("^", "**"), locals={n: sympy.Symbol(n) for n in names})
        except (sympy.SympifyError, SyntaxError, TypeError) as err:
            raise BCDataError(f"Could not evaluate BC expression `{text}` with signature {names}.") from err

    @property
    def value_cell_index(self) -> int:
        """Index (into the valid array along ``axis``) of the cell whose value enters the expression."""
        n = int(self.grid.shape[self.axis])
        if self.value_cell is None:
            return n - 1 if self.upper else 0
        return int(self.value_cell) % n

    def wall_coordinates(self) -> list[np.ndarray]:
        """Coordinates of the wall points of this face, one array of the face's shape per grid axis
        (``GridBase._boundary_coordinates``, pde/grids/base.py)."""
        grid = self.grid
        coords = []
        for a in range(grid.num_axes):
            if a == self.axis:
                coords.append(np.array([grid.axes_bounds[a][1] if self.upper else grid.axes_bounds[a][0]], dtype=np.float64))
            else:
                coords.append(np.asarray(grid.axes_coords[a], dtype=np.float64))
        mesh = np.meshgrid(*coords, indexing="ij")
        return [np.take(m, 0, axis=self.axis) for m in mesh]

    def get_virtual_point_data(self):
        msg = "expression boundary conditions have no constant virtual-point data"
        raise NotImplementedError(msg)

    def __repr__(self) -> str:
        side = "upper" if self.upper else "lower"
        return f"{self.__class__.__name__}(axis={self.axis}, {side}, {self.target}={self._input['value_expr']!r})"


class ExpressionValueBC(ExpressionBC):
    names = ["value_expression", "value_expr"]
    target = "value"


class ExpressionDerivativeBC(ExpressionBC):
    names = ["derivative_expression", "derivative_expr"]
    target = "derivative"


class ExpressionMixedBC(ExpressionBC):
    names = ["mixed_expression", "mixed_expr", "robin_expression", "robin_expr"]
    target = "mixed"


class NormalDirichletBC(DirichletBC):
    names = ["normal_value", "normal_dirichlet", "dirichlet_normal"]
    normal = True


class NormalNeumannBC(NeumannBC):
    names = ["normal_derivative", "normal_neumann", "neumann_normal"]
    normal = True


class NormalMixedBC(MixedBC):
    names = ["normal_mixed", "normal_robin"]
    normal = True


class NormalCurvatureBC(CurvatureBC):
    names = ["normal_curvature"]
    normal = True


class BoundaryPair:
    """Two independent conditions on one axis (axis.py:221-238)."""

    periodic = False

    def __init__(self, low: BCBase, high: BCBase):
        self.low, self.high = low, high
        self.grid, self.axis = low.grid, low.axis

    def __iter__(self):
        yield self.low
        yield self.high

    def __getitem__(self, index):
        return self.high if index in (1, True) else self.low

    def __repr__(self) -> str:
        return f"BoundaryPair({self.low!r}, {self.high!r})"


class BoundaryPeriodic(BoundaryPair):
    periodic = True

    def __init__(self, grid, axis: int, *, rank: int = 0, flip_sign: bool = False):
        super().__init__(
            _PeriodicBC(grid, axis, False, rank=rank, flip_sign=flip_sign),
            _PeriodicBC(grid, axis, True, rank=rank, flip_sign=flip_sign),
        )
        self.flip_sign = flip_sign


def _is_local_bc_data(data: dict) -> bool:
    return "type" in data or any(k in BCBase._conditions for k in data)


def get_boundary_axis(grid, axis: int, data, rank: int = 0) -> BoundaryPair:
    """Boundary condition of one axis from data (axis.py:392-452)."""
    if isinstance(data, (list, tuple)) and len(data) == 2:
        try:
            if data[0] == data[1]:
                data = data[0]
        except ValueError:
            pass
    if isinstance(data, str) and data.startswith("auto_periodic_"):
        data = "periodic" if grid.periodic[axis] else data[len("auto_periodic_") :]
    if isinstance(data, BoundaryPair):
        bcs = data
    elif isinstance(data, str) and data == "periodic" or (isinstance(data, dict) and data.get("type") == "periodic"):
        bcs = BoundaryPeriodic(grid, axis, rank=rank)
    elif isinstance(data, str) and data == "anti-periodic" or (isinstance(data, dict) and data.get("type") == "anti-periodic"):
        bcs = BoundaryPeriodic(grid, axis, rank=rank, flip_sign=True)
    elif isinstance(data, dict) and ("low" in data or "high" in data):
        low = BCBase.from_data(grid, axis, False, data["low"], rank=rank)
        high = BCBase.from_data(grid, axis, True, data["high"], rank=rank)
        bcs = BoundaryPair(low, high)
    elif isinstance(data, (str, dict, BCBase)):
        bcs = BoundaryPair(BCBase.from_data(grid, axis, False, data, rank=rank), BCBase.from_data(grid, axis, True, data, rank=rank))
    elif isinstance(data, (list, tuple)) and len(data) == 2:
        if any(isinstance(d, str) and d == "periodic" for d in data):
            msg = f"Only one side of {grid.axes[axis]} axis was set to have periodic boundary conditions."
            raise BCDataError(msg)
        bcs = BoundaryPair(BCBase.from_data(grid, axis, False, data[0], rank=rank), BCBase.from_data(grid, axis, True, data[1], rank=rank))
    else:
        raise BCDataError(f"Unsupported boundary format: `{data}`.")
    if bcs.periodic != grid.periodic[axis]:
        # axis.py: periodicity of grid and boundary condition must agree
        msg = f"Periodicity of conditions must match grid (axis {grid.axes[axis]}: grid periodic={grid.periodic[axis]})"
        raise RuntimeError(msg)
    return bcs


class BoundariesList:
    """Boundary conditions of all axes of a grid (axes.py:107-345)."""

    def __init__(self, boundaries: list[BoundaryPair]):
        self._axes = list(boundaries)
        self.grid = boundaries[0].grid
        if len(self._axes) != self.grid.num_axes:
            msg = f"Need boundary conditions for {self.grid.num_axes} axes"
            raise ValueError(msg)

    def __iter__(self):
        return iter(self._axes)

    def __len__(self) -> int:
        return len(self._axes)

    def __getitem__(self, index):
        return self._axes[index]

    def __repr__(self) -> str:
        return f"BoundariesList({self._axes!r})"

    @property
    def periodic(self) -> list[bool]:
        return [bc.periodic for bc in self._axes]

    @classmethod
    def from_data(cls, data, *, grid, rank: int = 0) -> "BoundariesList":
        if isinstance(data, BoundariesList):
            if data.grid != grid:
                msg = f"The grid of the supplied boundary condition is incompatible with the current grid ({data.grid!r} != {grid!r})"
                raise ValueError(msg)
            return data
        if callable(data):
            msg = "hip backend: ghost cells set by python callbacks are not supported"
            raise NotImplementedError(msg)
        if isinstance(data, str):
            return cls([get_boundary_axis(grid, i, data, rank=rank) for i in range(grid.num_axes)])
        if isinstance(data, dict):
            if _is_local_bc_data(data) or "low" in data or "high" in data:
                return cls([get_boundary_axis(grid, i, data, rank=rank) for i in range(grid.num_axes)])
            data = dict(data)
            bc_all = data.pop("*", None)
            per_axis: list[list[Any]] = [[bc_all, bc_all] for _ in range(grid.num_axes)]
            for ax, name in enumerate(grid.axes):
                if (both := data.pop(name, None)) is not None:
                    per_axis[ax] = [both, both]
                if (low := data.pop(name + "-", None)) is not None:
                    per_axis[ax][0] = low
                if (high := data.pop(name + "+", None)) is not None:
                    per_axis[ax][1] = high
            if data:
                raise BCDataError(f"Didn't use BC data from {list(data)}")
            missing = [grid.axes[ax] + "-+"[i] for ax in range(grid.num_axes) for i in range(2) if per_axis[ax][i] is None]
            if missing:
                raise BCDataError(f"Didn't specify BCs for {missing}")
            return cls([get_boundary_axis(grid, i, tuple(b), rank=rank) for i, b in enumerate(per_axis)])
        if hasattr(data, "__len__"):
            if len(data) == grid.num_axes:
                return cls([get_boundary_axis(grid, i, b, rank=rank) for i, b in enumerate(data)])
            if grid.num_axes == 1 and len(data) == 2:
                return cls([get_boundary_axis(grid, 0, data, rank=rank)])
            raise BCDataError(f"Got {len(data)} boundary conditions, but grid has {grid.num_axes} axes.")
        raise BCDataError(f"Unsupported boundary format: `{data}`.")
