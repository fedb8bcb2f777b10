"""Minimal Cartesian grids with the attribute names of ``pde.grids.cartesian``.

Only what the hot path needs is mirrored (``pde/grids/cartesian.py:36-146``, ``:473-507``;
``pde/grids/base.py:314-337``, ``:999-1034``, ``:1152-1261``): geometry, ``get_boundary_conditions``
and the ``make_operator`` / ``make_operator_no_bc`` dispatch into a backend.  A real py-pde
``CartesianGrid`` exposes the same attributes, so :class:`pde_hip.backend.HipBackend` accepts
either.
"""

from __future__ import annotations

import numpy as np

_AXES = "xyz"


class CartesianGrid:
    """d-dimensional Cartesian grid with uniform discretization per axis."""

    def __init__(self, bounds, shape, periodic=False):
        bounds = np.array(bounds, dtype=np.double, ndmin=1)
        if np.isscalar(shape):  # same number of cells along every axis (cartesian.py:100-110)
            shape = (int(shape),) * (1 if bounds.ndim == 1 else len(bounds))
        else:
            shape = tuple(int(s) for s in shape)
        if bounds.shape == (2,):
            bounds = np.broadcast_to(bounds, (len(shape), 2)).copy()
        if bounds.shape != (len(shape), 2):
            msg = f"Incompatible number of dimensions in bounds {bounds.shape} and shape {shape}"
            raise ValueError(msg)
        if not 1 <= len(shape) <= 3:
            msg = "hip mirror grids support 1 to 3 dimensions"
            raise NotImplementedError(msg)
        if any(s < 1 for s in shape):
            msg = "Grid shape must be positive"
            raise ValueError(msg)
        self._shape = shape
        self._bounds = tuple((float(lo), float(hi)) for lo, hi in bounds)
        if isinstance(periodic, (bool, np.bool_)):
            periodic = [bool(periodic)] * len(shape)
        elif len(periodic) != len(shape):
            msg = "Number of axes did not match number of periodic flags"
            raise ValueError(msg)
        self._periodic = [bool(p) for p in periodic]
        # dx = (hi - lo) / N, cell centres lo + (i + 1/2) dx   (cartesian.py:48-59, :126-136)
        self._discretization = np.array([(hi - lo) / n for (lo, hi), n in zip(self._bounds, shape)])
        self._mesh = None

    # geometry -----------------------------------------------------------------------------
    @property
    def dim(self) -> int:
        return len(self._shape)

    num_axes = dim

    @property
    def axes(self) -> list[str]:
        return list(_AXES[: self.dim])

    @property
    def shape(self) -> tuple[int, ...]:
        return self._shape

    @property
    def _shape_full(self) -> tuple[int, ...]:
        return tuple(n + 2 for n in self._shape)

    @property
    def periodic(self) -> list[bool]:
        return self._periodic

    @property
    def discretization(self) -> np.ndarray:
        return self._discretization

    @property
    def axes_bounds(self) -> tuple[tuple[float, float], ...]:
        return self._bounds

    @property
    def axes_coords(self) -> tuple[np.ndarray, ...]:
        return tuple(lo + (np.arange(n) + 0.5) * dx for (lo, _), n, dx in zip(self._bounds, self._shape, self._discretization))

    @property
    def cell_coords(self) -> np.ndarray:
        return np.moveaxis(np.array(np.meshgrid(*self.axes_coords, indexing="ij")), 0, -1)

    @property
    def cell_volumes(self) -> float:
        return float(np.prod(self._discretization))

    @property
    def volume(self) -> float:
        return float(np.prod([hi - lo for lo, hi in self._bounds]))

    @property
    def _idx_valid(self) -> tuple[slice, ...]:
        return (slice(1, -1),) * self.dim

    def __eq__(self, other) -> bool:
        return (
            isinstance(other, CartesianGrid)
            and self._shape == other._shape
            and self._bounds == other._bounds
            and self._periodic == other._periodic
        )

    def __hash__(self) -> int:
        return hash((self._shape, self._bounds, tuple(self._periodic)))

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(bounds={self._bounds}, shape={self._shape}, periodic={self._periodic})"

    # boundary conditions / operators ----------------------------------------------------------
    def get_boundary_conditions(self, bc="auto_periodic_neumann", rank: int = 0):
        """Parse ``bc`` into a :class:`~pde_hip.boundaries.BoundariesList` (grids/base.py:999-1034)."""
        from .boundaries import BoundariesList

        return BoundariesList.from_data(bc, grid=self, rank=rank)

    def make_operator_no_bc(self, operator, *, backend="hip", **kwargs):
        from .backend import get_backend

        return get_backend(backend).make_operator_no_bc(self, operator, **kwargs)

    def make_operator(self, operator, bc=None, *, backend="hip", dtype=None, **kwargs):
        """Operator with boundary conditions: ``op(arr, out=None, args=None)`` (grids/base.py:1198-1261)."""
        from .backend import get_backend

        backend_impl = get_backend(backend)
        info = backend_impl.get_operator_info(self, operator)
        bcs = self.get_boundary_conditions(bc if bc is not None else "auto_periodic_neumann", rank=info.rank_in)
        return backend_impl.make_operator(self, info, bcs=bcs, dtype=dtype, **kwargs)


class UnitGrid(CartesianGrid):
    """Grid with unit discretization, bounds [0, N] (cartesian.py:473-507)."""

    def __init__(self, shape, periodic=False):
        shape = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
        super().__init__([(0, n) for n in shape], shape, periodic)
