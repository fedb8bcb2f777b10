"""Slab decomposition of a Cartesian grid along axis 0 (one slab per GPU / process).

Partitioning arithmetic of the reference's ``GridMesh`` (``pde/grids/_mesh.py:96-111`` cell
ranges, ``:401-444`` neighbours incl. periodic wrap, ``:535-569`` inter-node faces replaced by an
exchange) restricted to the ``[n, 1, 1]`` decomposition: with axis-0 slabs in C order every halo
face is one contiguous block, and every rank has exactly two neighbours (2 of its 7 xGMI links).
The reference's "auto" choice would be 2x2x2 for 512^3 on 8 ranks — slabs are a deliberate
deviation.
"""

from __future__ import annotations

import numpy as np

def subdivide(num: int, chunks: int) -> np.ndarray:
    """Cells per chunk, identical to ``_subdivide`` in pde/grids/_mesh.py:96-111."""
    if chunks > num:
        msg = f"Cannot divide {num} cells into {chunks} slabs"
        raise RuntimeError(msg)
    return np.diff(np.linspace(0, num, chunks + 1).astype(int))


class SlabMesh:
    """The part of a grid owned by ``rank`` out of ``size`` ranks."""

    def __init__(self, grid: CartesianGrid, size: int, rank: int):
        if not 0 <= rank < size:
            msg = f"rank {rank} outside of world size {size}"
            raise ValueError(msg)
        self.grid, self.size, self.rank = grid, int(size), int(rank)
        counts = subdivide(grid.shape[0], size)
        offsets = np.concatenate([[0], np.cumsum(counts)])
        self.counts = counts
        self.lo, self.hi = int(offsets[rank]), int(offsets[rank + 1])
        self.n_local = self.hi - self.lo
        periodic0 = bool(grid.periodic[0])
        # neighbours along axis 0 (None = physical boundary)
        if size == 1:
            self.lower = self.upper = None
        else:
            self.lower = rank - 1 if rank > 0 else (size - 1 if periodic0 else None)
            self.upper = rank + 1 if rank < size - 1 else (0 if periodic0 else None)
        self.local_shape = (self.n_local, *[int(n) for n in grid.shape[1:]])
        self._subgrid = None

    @property
    def subgrid(self):
        """The slab as a grid object of the mirror classes (tests / mirror API only; the stepper itself needs just
        ``local_shape`` and the parent's discretization, so it also takes the real py-pde's grids)."""
        if self._subgrid is None:
            from .grids import CartesianGrid

            grid = self.grid
            bounds_all = getattr(grid, "axes_bounds", None)
            (lo_b, _), dx0 = bounds_all[0], grid.discretization[0]
            bounds = [(lo_b + self.lo * dx0, lo_b + self.hi * dx0), *bounds_all[1:]]
            periodic = [bool(grid.periodic[0]) and self.size == 1, *[bool(p) for p in grid.periodic[1:]]]
            self._subgrid = CartesianGrid(bounds, self.local_shape, periodic)
            # keep the discretization bit-identical to the parent grid (bounds arithmetic may round)
            self._subgrid._discretization = np.array(grid.discretization, dtype=float)
        return self._subgrid

    def slab_faces(self, bcs, *, force_exchange: bool = False, upload=None):
        """Face table of THIS slab from the boundary conditions of the WHOLE grid (any ``BoundariesList``: the mirror's or
        py-pde's own).  Replaces ``GridMesh.extract_boundary_conditions`` + ``_MPIBC`` (pde/grids/_mesh.py:535-569,
        pde/grids/boundaries/local.py:561-662): faces towards a neighbour are marked SKIP (their ghost layer is filled by the
        halo exchange), physical faces of axis 0 keep their condition with the index translated into the slab, per-face arrays
        of the other axes are sliced along axis 0 (the reference refuses those: ``Cannot transfer complicated BC to subgrid``,
        local.py:1515-1540)."""
        from . import _abi
        from .backend import FaceTable, _upload_f64, convert_bcs

        if upload is None:
            upload = _upload_f64

        class _Host:
            def __init__(self, arr):
                self.arr = np.ascontiguousarray(arr, dtype=np.float64)
                self.ptr = self.arr.ctypes.data

        for pair in bcs:
            for bc in (pair.low, pair.high):
                if getattr(bc, "rank", 0) != 0 or getattr(bc, "normal", False):
                    # the slab loops advance scalar fields (rank-0 conditions); `normal_*` conditions belong to vector fields
                    msg = "slab-parallel stepping supports conditions of scalar fields only (got a rank-1 / `normal` condition)"
                    raise NotImplementedError(msg)
        glob = convert_bcs(bcs, upload=_Host)
        by_ptr = {h.ptr: h.arr for h in glob.keepalive}
        exchanged = {(0, False): self.lower is not None, (0, True): self.upper is not None}
        if force_exchange and self.size == 1 and bool(self.grid.periodic[0]):
            exchanged = {(0, False): True, (0, True): True}
        out = FaceTable()
        nd = len(self.grid.shape)
        for ax in range(nd):
            for upper in (False, True):
                src, dst = glob.c[2 * ax + int(upper)], out.c[2 * ax + int(upper)]
                if ax == 0 and exchanged[(0, upper)]:
                    # the exchange copies the neighbour's layer as it is: only the plain periodic wrap-around may be replaced by
                    # it.  An anti-periodic axis (flip_sign: factor -1, pde/grids/boundaries/local.py:1728-1731) would silently
                    # run as periodic; the reference's `_MPIBC` is likewise only installed for periodic=True faces
                    wraps = self.rank == (self.size - 1 if upper else 0)     # the face of the WHOLE grid (inner faces have no condition)
                    if wraps and (src.kind != _abi.BC_ORDER1 or (src.flags & _abi.BCF_ARRAYS) or src.factor1 != 1.0 or src.const_v != 0.0):
                        msg = "anti-periodic axis 0 cannot be slab decomposed (the halo exchange copies the neighbour's layer unchanged)"
                        raise NotImplementedError(msg)
                    dst.kind = _abi.BC_SKIP
                    continue
                if ax == 0 and self.size > 1 and src.kind != _abi.BC_SKIP:
                    # a physical face of the decomposed axis: the virtual point reads a cell of THIS slab
                    for idx in ([src.index1] if src.kind == _abi.BC_ORDER1 else [src.index1, src.index2]):
                        if not self.lo <= idx < self.hi:
                            msg = "boundary condition of the decomposed axis reads a cell of another slab"
                            raise NotImplementedError(msg)
                dst.kind, dst.flags = src.kind, src.flags
                shift = self.lo if ax == 0 else 0
                dst.index1, dst.index2 = src.index1 - shift, (src.index2 - shift if src.kind == _abi.BC_ORDER2 else src.index2)
                dst.const_v, dst.factor1, dst.factor2 = src.const_v, src.factor1, src.factor2
                if src.flags & _abi.BCF_ARRAYS:
                    for name in ("const_arr", "factor1_arr", "factor2_arr"):
                        ptr = getattr(src, name)
                        if not ptr:
                            continue
                        arr = by_ptr[ptr]
                        if ax > 0:   # face arrays of axes >= 1 have axis 0 as their first face axis (after the tensor axes)
                            lead = arr.ndim - (nd - 1)
                            arr = arr[(slice(None),) * lead + (slice(self.lo, self.hi),)]
                        buf = upload(np.ascontiguousarray(arr))
                        out.keepalive.append(buf)
                        setattr(dst, name, buf.ptr)
        return out

    @property
    def exchanged_faces(self) -> set[tuple[int, bool]]:
        """(axis, upper) faces whose ghost layer comes from a neighbour instead of a BC."""
        faces = set()
        if self.lower is not None:
            faces.add((0, False))
        if self.upper is not None:
            faces.add((0, True))
        return faces

    def extract(self, data: np.ndarray) -> np.ndarray:
        """Local block of global valid data (``GridMesh.extract_field_data``, _mesh.py:446-479)."""
        nd = self.grid.num_axes
        idx = (...,) + (slice(self.lo, self.hi),) + (slice(None),) * (nd - 1)
        return data[idx]

    def sub_boundaries(self, bcs):
        """BCs of the slab: physical BCs stay on outer faces; inter-slab faces get a placeholder
        (they are listed in :attr:`exchanged_faces` and never evaluated).  Per-face arrays of the
        other axes are sliced along axis 0; the reference refuses those
        (``Cannot transfer complicated BC to subgrid``, local.py:1515-1540)."""
        from .boundaries import BoundariesList, BoundaryPair, BoundaryPeriodic, _PeriodicBC

        out = []
        for ax, pair in enumerate(bcs):
            if ax == 0:
                if isinstance(pair, BoundaryPeriodic) and self.size > 1:
                    low = _PeriodicBC(self.subgrid, 0, False, rank=pair.low.rank, flip_sign=pair.flip_sign)
                    high = _PeriodicBC(self.subgrid, 0, True, rank=pair.low.rank, flip_sign=pair.flip_sign)
                    if pair.flip_sign:
                        msg = "anti-periodic axis 0 cannot be slab decomposed"
                        raise NotImplementedError(msg)
                    out.append(BoundaryPair(low, high))
                    continue
                out.append(self._rebind_pair(pair, slice_axis0=False))
            else:
                out.append(self._rebind_pair(pair, slice_axis0=True))
        return BoundariesList(out)

    def _rebind_pair(self, pair, *, slice_axis0: bool):
        from .boundaries import BoundaryPair, BoundaryPeriodic

        if isinstance(pair, BoundaryPeriodic):
            return BoundaryPeriodic(self.subgrid, pair.axis, rank=pair.low.rank, flip_sign=pair.flip_sign)
        return BoundaryPair(self._rebind(pair.low, slice_axis0), self._rebind(pair.high, slice_axis0))

    def _rebind(self, bc, slice_axis0: bool):
        import copy

        new = copy.copy(bc)
        new.grid = self.subgrid
        if slice_axis0:
            # face arrays of axes >= 1 have axis 0 as their first face axis; value and const are sliced independently,
            # each only when it really is a per-face array (a MixedBC may carry a scalar value and an array const)
            lead = len(bc._shape_tensor)
            idx = (slice(None),) * lead + (slice(self.lo, self.hi),)
            if np.ndim(bc.value) > lead:
                new._value = np.ascontiguousarray(bc.value[idx])
            if hasattr(bc, "const") and np.ndim(bc.const) > lead:
                new.const = np.ascontiguousarray(bc.const[idx])
        return new


def combine(blocks: list[np.ndarray], num_axes: int) -> np.ndarray:
    """Concatenate per-rank valid blocks along axis 0 (``combine_field_data``, _mesh.py:657-696)."""
    axis = blocks[0].ndim - num_axes
    return np.concatenate(blocks, axis=axis)
