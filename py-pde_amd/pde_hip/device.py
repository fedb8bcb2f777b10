"""Device-resident field data of the hip backend.

:class:`DeviceArray` is the backend's *native array* (the ``TNativeArray`` of
``pde/backends/base.py:65``): one field (scalar, or ``ncomp`` components) in the ghost-padded
"full" layout on the GPU, see ``include/pdehip.h`` for the layout contract.  Keeping the native
representation ghost-padded removes the valid→full copy the reference performs on every
operator call (``pde/backends/numpy/backend.py:105-113``).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from ._lib import require_device


class DeviceBuffer:
    """Owning handle of a raw device allocation (freed on garbage collection)."""

    def __init__(self, nbytes: int):
        self._lib = require_device()
        p = C.c_void_p()
        self._lib.malloc(C.byref(p), int(nbytes))
        self.ptr = p.value
        self.nbytes = int(nbytes)

    def free(self) -> None:
        if getattr(self, "ptr", None):
            try:
                self._lib.free(self.ptr)
            except Exception:  # noqa: BLE001 - interpreter shutdown
                pass
            self.ptr = None

    def __del__(self):
        self.free()


class GridInfo:
    """POD description of a Cartesian grid (+ dtype) shared by all arrays living on it."""

    def __init__(self, shape, dx, dtype=np.float64):
        self.shape = tuple(int(s) for s in shape)
        self.dx = tuple(float(d) for d in dx)
        self.dtype = np.dtype(dtype)
        self.c = _abi.make_grid(self.shape, self.dx, self.dtype)
        lib = require_device()
        lay = (C.c_int64 * 8)()
        lib.layout(C.byref(self.c), lay)
        self.comp_elems = int(lay[2])
        self.slack = int(lay[6])
        self.layer_pitch = int(lay[7])  # elements of one layer along axis 0
        self.num_cells = int(np.prod(self.shape))

    @property
    def ref(self):
        return C.byref(self.c)

    def sub(self, n0: int) -> "GridInfo":
        """Same grid with a different extent along axis 0 (slab of layers)."""
        return GridInfo((n0, *self.shape[1:]), self.dx, self.dtype)

    def key(self):
        return (self.shape, self.dx, self.dtype.str)


class DeviceArray:
    """A field in the device "full" layout; ``comp_shape`` = leading tensor dimensions."""

    def __init__(self, info: GridInfo, comp_shape=(), *, buffer: DeviceBuffer | None = None, ptr: int | None = None, complex_pairs: bool = False):
        """``complex_pairs``: the array holds COMPLEX data as planar real components - the last tensor axis (length 2) is (real part,
        imaginary part), ``info.dtype`` the real type; ``set_valid`` / ``get_valid`` then take / hand out complex host arrays of shape
        ``comp_shape[:-1] + grid`` (pde_hip/complex_expr.py: every stencil of this path has real coefficients)."""
        self.info = info
        self.comp_shape = tuple(int(c) for c in comp_shape)
        self.complex_pairs = bool(complex_pairs)
        if self.complex_pairs and (not self.comp_shape or self.comp_shape[-1] != 2):
            msg = "complex data needs a last tensor axis of length 2 (real part, imaginary part)"
            raise ValueError(msg)
        self.ncomp = int(np.prod(self.comp_shape)) if self.comp_shape else 1
        self.itemsize = info.dtype.itemsize
        self.nbytes = (self.ncomp * info.comp_elems + info.slack) * self.itemsize
        if ptr is not None:
            self._buffer = buffer  # keeps the owner alive (may be None for foreign memory)
            self.ptr = int(ptr)
        else:
            self._buffer = buffer if buffer is not None else DeviceBuffer(self.nbytes)
            self.ptr = self._buffer.ptr

    # --- numpy-like metadata (valid data) -------------------------------------------
    @property
    def shape(self) -> tuple[int, ...]:
        return self.comp_shape + self.info.shape

    @property
    def dtype(self):
        return self.info.dtype

    @property
    def ndim(self) -> int:
        return len(self.shape)

    def __repr__(self) -> str:
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype}, ptr=0x{self.ptr:x})"

    def empty_like(self) -> "DeviceArray":
        return DeviceArray(self.info, self.comp_shape, complex_pairs=self.complex_pairs)

    # --- complex host data <-> planar real components ---------------------------------------------------------------
    @property
    def host_dtype(self):
        """dtype of the host arrays ``set_valid`` takes / ``get_valid`` returns."""
        if not self.complex_pairs:
            return self.info.dtype
        return np.dtype(np.complex128 if self.info.dtype == np.float64 else np.complex64)

    @property
    def host_shape(self) -> tuple[int, ...]:
        return (self.comp_shape[:-1] if self.complex_pairs else self.comp_shape) + self.info.shape

    def _to_planar(self, host: np.ndarray) -> np.ndarray:
        axis = len(self.comp_shape) - 1
        host = np.asarray(host)
        return np.ascontiguousarray(np.stack([host.real, host.imag if np.iscomplexobj(host) else np.zeros_like(host.real)], axis=axis), dtype=self.info.dtype)

    def component(self, index: int) -> "DeviceArray":
        """View of ``arr[index]`` (first tensor axis), sharing memory with ``self``."""
        if not self.comp_shape:
            msg = "scalar fields have no components"
            raise IndexError(msg)
        sub = self.comp_shape[1:]
        stride = (int(np.prod(sub)) if sub else 1) * self.info.comp_elems * self.itemsize
        return DeviceArray(self.info, sub, buffer=self._buffer, ptr=self.ptr + int(index) * stride, complex_pairs=self.complex_pairs and len(sub) >= 1)

    def flat(self) -> "DeviceArray":
        """The same memory with ONE tensor axis: all components in C order (a rank-2 field as ``dim * dim`` scalar components)."""
        if len(self.comp_shape) <= 1:
            return self
        return DeviceArray(self.info, (self.ncomp,), buffer=self._buffer, ptr=self.ptr)

    def layer_ptr(self, layer: int, comp: int = 0) -> int:
        """Device address of full layer ``layer`` (0 = lower ghost layer) along axis 0."""
        return self.ptr + (comp * self.info.comp_elems + layer * self.info.layer_pitch) * self.itemsize

    # --- host <-> device ------------------------------------------------------------------
    def _host_strides(self, host: np.ndarray):
        """``host_strides[4]`` of pdehip_upload_valid / pdehip_download_valid for a host array of this array's shape, or
        None when its memory cannot be described that way (fastest axis not contiguous, tensor axes not collapsible)."""
        nd = len(self.info.shape)
        st = host.strides
        if host.shape != self.shape or host.dtype != self.dtype or not host.flags.aligned:
            return None
        if host.shape[-1] > 1 and st[-1] != self.itemsize:
            return None
        comp = 0
        ncs = len(self.comp_shape)
        if self.ncomp > 1:
            # the tensor axes must walk through memory like one axis of `ncomp` entries
            comp = st[ncs - 1]
            for a in range(ncs - 1):
                if self.comp_shape[a] > 1 and st[a] != st[a + 1] * self.comp_shape[a + 1]:
                    return None
        out = (C.c_int64 * 4)(comp, 0, 0, self.itemsize)
        for a in range(nd - 1):
            out[1 + 3 - nd + a] = st[ncs + a]
        return out

    def set_valid(self, data: np.ndarray, stream=None) -> "DeviceArray":
        """Upload valid data (host numpy, any strides) into the interior.  A view with a contiguous fastest axis - such
        as ``field.data`` of the reference, a window of the ghost-padded host array - is read in place."""
        lib = require_device()
        host = np.asarray(data)
        if self.complex_pairs:
            if host.shape != self.host_shape:
                msg = f"Incompatible shapes {host.shape} != {self.host_shape}"
                raise ValueError(msg)
            host = self._to_planar(host)
        if host.shape != self.shape:
            msg = f"Incompatible shapes {host.shape} != {self.shape}"
            raise ValueError(msg)
        strides = self._host_strides(host)
        if strides is None:
            host = np.ascontiguousarray(host, dtype=self.dtype)
            strides = self._host_strides(host)
        lib.upload_valid(self.info.ref, self.ncomp, host.ctypes.data, strides, self.ptr, stream)
        return self

    def set_hostfull(self, data_full: np.ndarray, stream=None) -> "DeviceArray":
        """Upload a reference-layout full array (ghost cells included)."""
        lib = require_device()
        host = np.ascontiguousarray(data_full, dtype=self.dtype)
        expect = self.comp_shape + tuple(s + 2 for s in self.info.shape)
        if host.shape != expect:
            msg = f"Incompatible shapes {host.shape} != {expect}"
            raise ValueError(msg)
        stage = DeviceBuffer(host.nbytes)
        lib.memcpy_h2d(stage.ptr, host.ctypes.data, host.nbytes, stream)
        lib.hostfull_to_full(self.info.ref, self.ncomp, stage.ptr, self.ptr, stream)
        lib.stream_synchronize(stream)
        stage.free()
        return self

    def get_valid(self, out: np.ndarray | None = None, stream=None) -> np.ndarray:
        """Download the interior as a host numpy array (written into ``out`` if given: in place when ``out`` has this
        array's dtype and a contiguous fastest axis, e.g. ``field.data`` of the reference)."""
        lib = require_device()
        if self.complex_pairs:
            planar = np.empty(self.shape, dtype=self.dtype)
            lib.download_valid(self.info.ref, self.ncomp, self.ptr, planar.ctypes.data, self._host_strides(planar), stream)
            axis = len(self.comp_shape) - 1
            res = np.take(planar, 0, axis=axis) + 1j * np.take(planar, 1, axis=axis)
            if out is not None:
                out[...] = res
                return out
            return res.astype(self.host_dtype, copy=False)
        strides = None
        if out is not None and out.flags.writeable:
            strides = self._host_strides(out)
        host = out if strides is not None else np.empty(self.shape, dtype=self.dtype)
        if strides is None:
            strides = self._host_strides(host)
        lib.download_valid(self.info.ref, self.ncomp, self.ptr, host.ctypes.data, strides, stream)
        if out is not None and host is not out:
            out[...] = host
            return out
        return host

    def get_hostfull(self, stream=None) -> np.ndarray:
        """Download including ghost cells, in the reference's compact full layout."""
        lib = require_device()
        host = np.empty(self.comp_shape + tuple(s + 2 for s in self.info.shape), dtype=self.dtype)
        stage = DeviceBuffer(host.nbytes)
        lib.full_to_hostfull(self.info.ref, self.ncomp, self.ptr, stage.ptr, stream)
        lib.memcpy_d2h(host.ctypes.data, stage.ptr, host.nbytes, stream)
        stage.free()
        return host

    def copy(self, stream=None) -> "DeviceArray":
        out = self.empty_like()
        require_device().memcpy_d2d(out.ptr, self.ptr, self.nbytes, stream)
        return out

    def __array__(self, dtype=None, copy=None):
        arr = self.get_valid()
        return arr.astype(dtype) if dtype is not None else arr


class DeviceScalar:
    """One fp64 value on the device (error norms of the adaptive steppers)."""

    def __init__(self):
        self._buf = DeviceBuffer(8)
        self.ptr = self._buf.ptr
        self._host = C.c_double(0.0)

    def value(self, stream=None) -> float:
        require_device().memcpy_d2h(C.addressof(self._host), self.ptr, 8, stream)
        return self._host.value


def ptr_array(arrays) -> C.Array:
    """void*[] of device addresses for the ``*_host`` pointer-table arguments."""
    arr = (C.c_void_p * len(arrays))()
    for i, a in enumerate(arrays):
        arr[i] = a.ptr if hasattr(a, "ptr") else int(a)
    return arr
