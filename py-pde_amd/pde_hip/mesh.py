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

    def slab_faces(self, bcs, *, force_exchange: bool = False, upload=None, comp_shape: tuple[int, ...] = (), component=None, part=None):
        """Face table of THIS slab from the boundary conditions of the WHOLE grid (any ``BoundariesList``: the mirror's or
        py-pde's own).  Replaces ``GridMesh.extract_boundary_conditions`` + ``_MPIBC`` (pde/grids/_mesh.py:535-569,
        pde/grids/boundaries/local.py:561-662): faces towards a neighbour are marked SKIP (their ghost layer is filled by the
        halo exchange), physical faces of axis 0 keep their condition with the index translated into the slab, per-face arrays
        of the other axes are sliced along axis 0 (the reference refuses those: ``Cannot transfer complicated BC to subgrid``,
        local.py:1515-1540).  ``comp_shape`` / ``component``: the conditions of a vector / tensor field, reduced to the table of ONE
        component for a scalar array (``convert_bcs``: the terms of vector operators inside expression PDEs)."""
        from . import _abi
        from .backend import FaceTable, _upload_f64, convert_bcs
        from .bc_expr import ExprFaceTable, _write_buffer, expression_faces, lower_expression_face

        if upload is None:
            upload = _upload_f64

        class _Host:
            def __init__(self, arr):
                self.arr = np.ascontiguousarray(arr, dtype=np.float64)
                self.ptr = self.arr.ctypes.data

        for pair in bcs:
            for bc in (pair.low, pair.high):
                if getattr(bc, "rank", 0) != (len(comp_shape) if component is not None else 0) or getattr(bc, "normal", False):
                    # the slab loops advance scalar fields (rank-0 conditions); `normal_*` conditions belong to vector fields
                    msg = "slab-parallel stepping supports conditions of scalar fields only (got a rank-1 / `normal` condition)"
                    raise NotImplementedError(msg)
        # conditions given as expressions (incl. time-dependent ones and ones that read the field): evaluated for THIS slab - the
        # face cut to the slab's layers, wall coordinates of the whole grid - and refreshed by a device program (bc_expr.py)
        expr_faces = expression_faces(bcs)
        if part is not None and expr_faces:
            # (single device: pde_hip/bc_expr.py splits `A + B * value`; the per-slab device programs have no complex form yet)
            msg = "hip backend: conditions given as expressions for complex fields on decomposed grids are not supported"
            raise NotImplementedError(msg)
        glob = convert_bcs(bcs, comp_shape if component is not None else (), skip=set(expr_faces), upload=_Host, component=component, part=part)
        by_ptr = {h.ptr: h.arr for h in glob.keepalive}
        exchanged = {(0, False): self.lower is not None, (0, True): self.upper is not None}
        if force_exchange and self.size == 1 and bool(self.grid.periodic[0]):
            exchanged = {(0, False): True, (0, True): True}
        out = FaceTable()
        dynamic = []
        nd = len(self.grid.shape)
        window = ([self.lo, *[0] * (nd - 1)], [self.hi, *[int(n) for n in self.grid.shape[1:]]])
        for ax in range(nd):
            for upper in (False, True):
                src, dst = glob.c[2 * ax + int(upper)], out.c[2 * ax + int(upper)]
                if (ax, upper) in expr_faces:
                    if ax == 0 and exchanged[(0, upper)]:
                        dst.kind = _abi.BC_SKIP          # an inner face: the wall of the whole grid belongs to another slab
                        continue
                    entry = lower_expression_face(expr_faces[(ax, upper)], out, upload, window)
                    if entry is not None:
                        dynamic.append(entry)
                    continue
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
        return ExprFaceTable(out, dynamic, _write_buffer) if expr_faces else out

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


# ---------------------------------------------------------------------------------------------
# block decomposition (multi-axis): the reference's GridMesh with its "auto" rule
# ---------------------------------------------------------------------------------------------
def optimal_decomposition(shape, size: int) -> list[int]:
    """Blocks per axis for ``size`` ranks: the rule of ``_get_optimal_decomposition`` (pde/grids/_mesh.py:59-93) - axes are
    visited from the shortest to the longest, each gets as many cuts as keep the blocks close to cubes.  512^3 on 8 ranks ->
    [2, 2, 2]."""
    shape_arr = np.asarray(shape)
    decomposition = [-1] * len(shape)
    order = np.argsort(shape_arr, kind="stable")
    size_left = int(size)
    for dim_count, dim in enumerate(order):
        shape_left = shape_arr[order[dim_count:]]
        node_size_estimate = int(np.prod(shape_left)) // size_left
        node_axis_len = max(1, int(np.floor(node_size_estimate ** (1 / len(shape_left)))))
        decomposition[dim] = int(np.clip(shape[dim] // node_axis_len, 1, size_left))
        size_left //= decomposition[dim]
    assert int(np.prod(decomposition)) <= size
    return decomposition


def block_decomposition(shape, size: int) -> list[int]:
    """Decomposition that uses ALL ``size`` ranks: the reference's rule where its product equals ``size`` (it raises
    "Node count incompatible with decomposition" otherwise, pde/grids/_mesh.py:251-256); else the factorisation of ``size`` with
    the smallest exchanged surface per block."""
    dims = optimal_decomposition(shape, size)
    if int(np.prod(dims)) == size:
        return dims
    nd = len(shape)
    best, best_cost = None, None

    def search(axis: int, left: int, cur: list[int]):
        nonlocal best, best_cost
        if axis == nd - 1:
            cand = [*cur, left]
            if any(d > n for d, n in zip(cand, shape)):
                return
            local = [n / d for n, d in zip(shape, cand)]
            cost = sum(np.prod([local[b] for b in range(nd) if b != a]) for a in range(nd) if cand[a] > 1)
            if best_cost is None or cost < best_cost:
                best, best_cost = cand, cost
            return
        for d in range(1, left + 1):
            if left % d == 0:
                search(axis + 1, left // d, [*cur, d])

    search(0, int(size), [])
    if best is None:
        msg = f"cannot cut a grid of shape {tuple(shape)} into {size} blocks"
        raise RuntimeError(msg)
    return best


class BlockMesh:
    """The box of a grid owned by ``rank`` in a ``dims[0] x dims[1] (x dims[2])`` decomposition (ranks in C order over the block
    indices, like the node ids of ``GridMesh``); cells per block by the reference's partition rule (``subdivide``,
    pde/grids/_mesh.py:96-111), neighbours incl. the periodic wrap (:401-444).  An axis with ONE block has no neighbours:
    its faces keep their conditions, periodic ones included."""

    def __init__(self, grid: CartesianGrid, dims, rank: int, *, force_exchange: bool = False):
        """``force_exchange``: a periodic axis with ONE block exchanges with the block itself instead of keeping its periodic
        condition (the pack / send / receive / unpack path on a single GPU: overhead probe and test of the device layer)."""
        nd = len(grid.shape)
        dims = [int(d) for d in dims]
        if len(dims) != nd or any(d < 1 for d in dims):
            msg = f"decomposition {dims} does not fit a grid with {nd} axes"
            raise ValueError(msg)
        self.grid, self.dims, self.size, self.rank = grid, dims, int(np.prod(dims)), int(rank)
        if not 0 <= rank < self.size:
            msg = f"rank {rank} outside of the {self.size} blocks of decomposition {dims}"
            raise ValueError(msg)
        self.index = [int(i) for i in np.unravel_index(rank, dims)]
        self.lo, self.hi, self.counts = [], [], []
        for a in range(nd):
            counts = subdivide(grid.shape[a], dims[a])
            offsets = np.concatenate([[0], np.cumsum(counts)])
            self.counts.append(counts)
            self.lo.append(int(offsets[self.index[a]]))
            self.hi.append(int(offsets[self.index[a] + 1]))
        self.local_shape = tuple(h - lo for lo, h in zip(self.lo, self.hi))
        # neighbour ranks per (axis, side); None = physical face
        self.neighbours: list[list[int | None]] = []
        for a in range(nd):
            pair: list[int | None] = [None, None]
            if dims[a] > 1 or (force_exchange and grid.periodic[a]):
                for side, step in ((0, -1), (1, 1)):
                    j = self.index[a] + step
                    if 0 <= j < dims[a] or grid.periodic[a]:
                        idx = list(self.index)
                        idx[a] = j % dims[a]
                        pair[side] = int(np.ravel_multi_index(idx, dims))
            self.neighbours.append(pair)

    @property
    def nb6(self):
        """``nb6[2 * axis + side]`` of ``pdehip_block_run`` (-1: physical face)."""
        import ctypes as C

        arr = (C.c_int * 6)(*([-1] * 6))
        for a, pair in enumerate(self.neighbours):
            for side in range(2):
                arr[2 * a + side] = -1 if pair[side] is None else int(pair[side])
        return arr

    def extract(self, data: np.ndarray) -> np.ndarray:
        """Local block of global valid data (``GridMesh.extract_field_data``, _mesh.py:446-479)."""
        return data[(...,) + tuple(slice(lo, hi) for lo, hi in zip(self.lo, self.hi))]

    def block_faces(self, bcs, *, upload=None, comp_shape: tuple[int, ...] = (), component=None, part=None):
        """Face table of THIS block from the conditions of the WHOLE grid: faces towards a neighbour are SKIP (filled by the
        exchange; the wrap-around of a decomposed periodic axis must be plainly periodic), physical faces keep their condition
        with the index translated into the block, per-face arrays are cut to the block's extent along the other axes."""
        from . import _abi
        from .backend import FaceTable, _upload_f64, convert_bcs
        from .bc_expr import ExprFaceTable, _write_buffer, expression_faces, lower_expression_face

        if upload is None:
            upload = _upload_f64

        class _Host:
            def __init__(self, arr):
                self.arr = np.ascontiguousarray(arr, dtype=np.float64)
                self.ptr = self.arr.ctypes.data

        for pair in bcs:
            for bc in (pair.low, pair.high):
                if getattr(bc, "rank", 0) != (len(comp_shape) if component is not None else 0) or getattr(bc, "normal", False):
                    msg = "block-parallel stepping supports conditions of scalar fields only"
                    raise NotImplementedError(msg)
        expr_faces = expression_faces(bcs)       # as in SlabMesh.slab_faces: cut to the block, coordinates of the whole grid
        if part is not None and expr_faces:
            # (single device: pde_hip/bc_expr.py splits `A + B * value`; the per-slab device programs have no complex form yet)
            msg = "hip backend: conditions given as expressions for complex fields on decomposed grids are not supported"
            raise NotImplementedError(msg)
        glob = convert_bcs(bcs, comp_shape if component is not None else (), skip=set(expr_faces), upload=_Host, component=component, part=part)
        by_ptr = {h.ptr: h.arr for h in glob.keepalive}
        out = FaceTable()
        dynamic = []
        nd = len(self.grid.shape)
        for ax in range(nd):
            for side, upper in enumerate((False, True)):
                src, dst = glob.c[2 * ax + side], out.c[2 * ax + side]
                if (ax, upper) in expr_faces:
                    if self.neighbours[ax][side] is not None:
                        dst.kind = _abi.BC_SKIP
                        continue
                    entry = lower_expression_face(expr_faces[(ax, upper)], out, upload, (self.lo, self.hi))
                    if entry is not None:
                        dynamic.append(entry)
                    continue
                if self.neighbours[ax][side] is not None:
                    wraps = self.index[ax] == (self.dims[ax] - 1 if upper else 0)
                    if wraps and (src.kind != _abi.BC_ORDER1 or (src.flags & _abi.BCF_ARRAYS) or src.factor1 != 1.0 or src.const_v != 0.0):
                        msg = f"anti-periodic axis {ax} cannot be decomposed (the exchange copies the neighbour's layer unchanged)"
                        raise NotImplementedError(msg)
                    dst.kind = _abi.BC_SKIP
                    continue
                if self.dims[ax] > 1 and src.kind != _abi.BC_SKIP:
                    for idx in ([src.index1] if src.kind == _abi.BC_ORDER1 else [src.index1, src.index2]):
                        if not self.lo[ax] <= idx < self.hi[ax]:
                            msg = "boundary condition of a decomposed axis reads a cell of another block"
                            raise NotImplementedError(msg)
                dst.kind, dst.flags = src.kind, src.flags
                shift = self.lo[ax]
                dst.index1, dst.index2 = src.index1 - shift, (src.index2 - shift if src.kind == _abi.BC_ORDER2 else src.index2)
                dst.const_v, dst.factor1, dst.factor2 = src.const_v, src.factor1, src.factor2
                if src.flags & _abi.BCF_ARRAYS:
                    others = [a for a in range(nd) if a != ax]
                    for name in ("const_arr", "factor1_arr", "factor2_arr"):
                        ptr = getattr(src, name)
                        if not ptr:
                            continue
                        arr = by_ptr[ptr]
                        lead = arr.ndim - (nd - 1)
                        cut = (slice(None),) * lead + tuple(slice(self.lo[a], self.hi[a]) for a in others)
                        buf = upload(np.ascontiguousarray(arr[cut]))
                        out.keepalive.append(buf)
                        setattr(dst, name, buf.ptr)
        return ExprFaceTable(out, dynamic, _write_buffer) if expr_faces else out


def combine_blocks(blocks: list[np.ndarray], dims, num_axes: int) -> np.ndarray:
    """Assemble per-rank valid blocks (rank order = C order of the block indices) into the global array
    (``combine_field_data``, _mesh.py:657-696)."""
    dims = [int(d) for d in dims]
    lead = blocks[0].ndim - num_axes
    nested = np.empty(dims, dtype=object)
    for r, b in enumerate(blocks):
        nested[np.unravel_index(r, dims)] = b

    def join(sub, axis):
        if axis == len(dims) - 1:
            return np.concatenate(list(sub), axis=lead + axis)
        return np.concatenate([join(s, axis + 1) for s in sub], axis=lead + axis)

    return join(nested, 0)
