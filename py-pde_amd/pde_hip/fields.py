"""Minimal ``ScalarField`` / ``VectorField`` / ``Tensor2Field`` with the reference's memory contract.

Mirrors what the hot path touches of ``pde.fields`` (``pde/fields/base.py:95-160``,
``pde/fields/datafield_base.py:93-201``, ``:827-963``, ``pde/fields/scalar.py:198-267``): a field
owns a C-contiguous *full* array with one ghost layer per side; ``.data`` is the strided interior
view; ``apply_operator`` / ``laplace`` / ``gradient`` / ``divergence`` dispatch to a backend.
"""

from __future__ import annotations

import numpy as np


class DataFieldBase:
    rank = 0

    def __init__(self, grid, data="zeros", *, label: str | None = None, dtype=None, with_ghost_cells: bool = False):
        self.grid = grid
        self.label = label
        shape_full = (grid.dim,) * self.rank + grid._shape_full
        if isinstance(data, DataFieldBase):
            data = data.data
        if isinstance(data, str):
            if data not in {"zeros", "empty", "ones"}:
                msg = f"Unknown data initialisation `{data}`"
                raise ValueError(msg)
            dt = np.dtype(dtype or np.double)
            self._data_full = np.ones(shape_full, dt) if data == "ones" else np.zeros(shape_full, dt)
        elif with_ghost_cells:
            arr = np.array(data, dtype=dtype, copy=True)
            if arr.shape != shape_full:
                msg = f"Incompatible shapes {arr.shape} != {shape_full}"
                raise ValueError(msg)
            self._data_full = np.ascontiguousarray(arr)
        else:
            arr = np.asarray(data, dtype=dtype)
            if not (np.issubdtype(arr.dtype, np.floating) or np.issubdtype(arr.dtype, np.complexfloating)):
                arr = arr.astype(np.double)
            self._data_full = np.zeros(shape_full, arr.dtype)
            self.data = np.broadcast_to(arr, (grid.dim,) * self.rank + grid.shape)

    # --- memory contract ------------------------------------------------------------------------
    @property
    def _idx_valid(self):
        return (...,) + (slice(1, -1),) * self.grid.num_axes

    @property
    def data(self) -> np.ndarray:
        """Interior view of the full array (fields/base.py:116-120)."""
        return self._data_full[self._idx_valid]

    @data.setter
    def data(self, value) -> None:
        self._data_full[self._idx_valid] = value

    @property
    def dtype(self):
        return self._data_full.dtype

    def copy(self, *, label=None, dtype=None):
        # ONE pass over the data (a 1 GB field: the second copy of `np.array` inside the constructor cost as much as the upload)
        new = self.__class__.__new__(self.__class__)
        new.grid, new.label = self.grid, label or self.label
        new._data_full = np.array(self._data_full, dtype=dtype or self.dtype, copy=True, order="C")
        return new

    @classmethod
    def random_uniform(cls, grid, vmin: float = 0, vmax: float = 1, *, label=None, dtype=None, rng=None):
        """U[vmin, vmax) samples (fields/datafield_base.py:150-201)."""
        rng = np.random.default_rng(rng)
        shape = (grid.dim,) * cls.rank + grid.shape
        data = rng.uniform(vmin, vmax, size=shape)
        return cls(grid, data, label=label, dtype=dtype)

    @classmethod
    def from_expression(cls, grid, expression: str, *, label=None, dtype=None):
        import sympy

        syms = [sympy.Symbol(a) for a in grid.axes]
        func = sympy.lambdify(syms, sympy.sympify(expression), modules="numpy")
        coords = [grid.cell_coords[..., i] for i in range(grid.dim)]
        data = np.broadcast_to(np.asarray(func(*coords), dtype=np.double), grid.shape)
        return cls(grid, np.array(data), label=label, dtype=dtype)

    # --- operators ------------------------------------------------------------------------------------
    def set_ghost_cells(self, bc, *, args=None, backend="hip") -> None:
        """Set the ghost cells of the host array (through the device ghost-cell kernel)."""
        from .backend import get_backend
        from .device import DeviceArray

        b = get_backend(backend)
        bcs = self.grid.get_boundary_conditions(bc, rank=self.rank)
        dev = DeviceArray(b.grid_info(self.grid, self.dtype), (self.grid.dim,) * self.rank)
        dev.set_hostfull(self._data_full, b.stream)
        b.make_ghost_cell_setter(bcs)(dev, args=args)
        self._data_full[...] = dev.get_hostfull(b.stream)

    def apply_operator(self, operator: str, bc, out=None, *, label=None, args=None, backend="hip", **kwargs):
        """Apply a (differential) operator with BCs (fields/datafield_base.py:900-963)."""
        from .backend import get_backend

        b = get_backend(backend)
        info = b.get_operator_info(self.grid, operator)
        if info.rank_in != self.rank:
            msg = f"Operator {operator} needs a field of rank {info.rank_in}"
            raise TypeError(msg)
        out_cls = {0: ScalarField, 1: VectorField, 2: Tensor2Field}[info.rank_out]
        if out is None:
            out = out_cls(self.grid, "empty", label=label, dtype=self.dtype)
        elif not isinstance(out, out_cls):
            msg = f"`out` must be a {out_cls.__name__}"
            raise TypeError(msg)
        if bc is None:
            op = b.make_operator_no_bc(self.grid, info, **kwargs)
            b._apply_operator(op, self._data_full, out=out.data, grid=self.grid)
        else:
            bcs = self.grid.get_boundary_conditions(bc, rank=self.rank)
            op = b.make_operator(self.grid, info, bcs=bcs, dtype=self.dtype, **kwargs)
            op(self.data, out=out.data, args=args)
        return out


class ScalarField(DataFieldBase):
    rank = 0

    def laplace(self, bc, out=None, **kwargs) -> "ScalarField":
        return self.apply_operator("laplace", bc=bc, out=out, **kwargs)

    def gradient(self, bc, out=None, **kwargs) -> "VectorField":
        return self.apply_operator("gradient", bc=bc, out=out, **kwargs)

    def gradient_squared(self, bc, out=None, **kwargs) -> "ScalarField":
        return self.apply_operator("gradient_squared", bc=bc, out=out, **kwargs)


class VectorField(DataFieldBase):
    rank = 1

    def divergence(self, bc, out=None, **kwargs) -> ScalarField:
        return self.apply_operator("divergence", bc=bc, out=out, **kwargs)

    def gradient(self, bc, out=None, **kwargs) -> "Tensor2Field":
        return self.apply_operator("vector_gradient", bc=bc, out=out, **kwargs)

    def laplace(self, bc, out=None, **kwargs) -> "VectorField":
        return self.apply_operator("vector_laplace", bc=bc, out=out, **kwargs)


class Tensor2Field(DataFieldBase):
    rank = 2

    def divergence(self, bc, out=None, **kwargs) -> VectorField:
        return self.apply_operator("tensor_divergence", bc=bc, out=out, **kwargs)


class FieldCollection:
    """Several fields on one grid stored as ONE array with a leading axis over all their components (``pde/fields/collection.py``):
    ``data`` has shape ``(n, *grid.shape)`` with ``n`` = the sum of the fields' component counts (a scalar 1, a vector ``dim``, a
    rank-2 field ``dim * dim`` in C order), ``collection[i]`` is a field of its own class viewing the same memory.  Just enough
    of the reference class for multi-field expression PDEs on the GPU box (which has no py-pde)."""

    def __init__(self, fields, *, copy_fields: bool = True, dtype=None):
        fields = list(fields)
        if not fields or any(not isinstance(f, DataFieldBase) for f in fields):
            msg = "FieldCollection (mirror) holds scalar, vector and rank-2 tensor fields"
            raise NotImplementedError(msg)
        self.grid = fields[0].grid
        dt = np.dtype(dtype or np.result_type(*[f.dtype for f in fields]))
        dim = self.grid.num_axes
        counts = [dim ** f.rank for f in fields]
        self._data_full = np.zeros((sum(counts),) + self.grid._shape_full, dt)
        self._fields = []
        start = 0
        for f, c in zip(fields, counts):
            block = self._data_full[start:start + c]
            start += c
            view = type(f).__new__(type(f))
            view.grid, view.label = self.grid, f.label
            view._data_full = block[0] if f.rank == 0 else block.reshape((dim,) * f.rank + self.grid._shape_full)
            view._data_full[...] = f._data_full
            self._fields.append(view)

    @property
    def data(self) -> np.ndarray:
        return self._data_full[(slice(None),) + (slice(1, -1),) * self.grid.num_axes]

    @data.setter
    def data(self, value) -> None:
        self.data[...] = value

    @property
    def dtype(self):
        return self._data_full.dtype

    def __iter__(self):
        return iter(self._fields)

    def __len__(self) -> int:
        return len(self._fields)

    def __getitem__(self, index):
        return self._fields[index]

    def copy(self, *, label=None, dtype=None):
        return FieldCollection([f.copy() for f in self._fields], dtype=dtype or self.dtype)
