"""Host <-> device transfers of valid data (`pdehip_upload_valid` / `pdehip_download_valid`, include/pdehip.h).

The reference keeps a field in host memory with its ghost cells; `field.data` is a strided window of that array
(pde/fields/base.py:116-160).  The transfers read / write such windows in place, through pinned chunks (several threads for
large fields).  The same checks run against the tests-only host shim (CPU, host logic + stride arithmetic) and against the HIP
library (`-m gpu`: the chunked pinned pipeline itself, including sizes that span many chunks and all copy threads).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

import pde_hip
from pde_hip.device import DeviceArray, GridInfo

CASES = [
    ((37,), (), np.float64),
    ((19, 23), (), np.float64),
    ((19, 23), (2,), np.float32),
    ((5, 6, 7), (), np.float64),
    ((5, 6, 7), (3,), np.float64),
    ((4, 3, 9), (3, 3), np.float32),
]


def _window(rng, shape, comp, dtype):
    """A `field.data`-like window: the interior of a ghost-padded host array."""
    full = rng.random(comp + tuple(s + 2 for s in shape)).astype(dtype)
    view = full[(Ellipsis,) + tuple(slice(1, -1) for _ in shape)]
    return full, view


def _roundtrip(shape, comp, dtype, rng):
    info = GridInfo(shape, (1.0,) * len(shape), np.dtype(dtype))
    full, view = _window(rng, shape, comp, dtype)
    assert not view.flags.c_contiguous or len(shape) == 1
    dev = DeviceArray(info, comp).set_valid(view)
    # contiguous download
    np.testing.assert_array_equal(dev.get_valid(), view)
    # download into another window: only the interior of the target changes
    target_full = np.full_like(full, -7)
    target = target_full[(Ellipsis,) + tuple(slice(1, -1) for _ in shape)]
    got = dev.get_valid(out=target)
    assert got is target
    np.testing.assert_array_equal(target, view)
    target[...] = -7
    assert (target_full == -7).all()        # ghost cells of the host array untouched
    # views the C side cannot address (fastest axis strided, other dtype) go through a contiguous copy
    rev = np.ascontiguousarray(view[..., ::-1])
    dev.set_valid(rev[..., ::-1])
    np.testing.assert_array_equal(dev.get_valid(), view)
    out64 = np.zeros(comp + tuple(shape), dtype=np.float64 if dtype == np.float32 else np.float32)
    dev.get_valid(out=out64)
    np.testing.assert_array_equal(out64, view.astype(out64.dtype))
    # Fortran-ordered tensor axes are not collapsible -> copy path, same result
    if len(comp) == 2:
        weird = np.asfortranarray(np.ascontiguousarray(view))
        dev.set_valid(weird)
        np.testing.assert_array_equal(dev.get_valid(), view)


@pytest.mark.parametrize("shape,comp,dtype", CASES)
def test_transfer_windows_shim(shape, comp, dtype):
    import shimlib

    with shimlib.use_shim():
        _roundtrip(shape, comp, dtype, np.random.default_rng(3))


def test_transfer_errors_shim():
    import shimlib

    with shimlib.use_shim() as lib:
        _errors(lib)


def _errors(lib):
    info = GridInfo((4, 5), (1.0, 1.0), np.dtype(np.float64))
    dev = DeviceArray(info)
    host = np.zeros((4, 5))
    bad = (C.c_int64 * 4)(0, 0, 40, 16)   # fastest axis not contiguous
    with pytest.raises(ValueError, match="contiguous along the fastest axis"):
        lib.upload_valid(info.ref, 1, host.ctypes.data, bad, dev.ptr, None)
    with pytest.raises(ValueError, match="contiguous along the fastest axis"):
        lib.download_valid(info.ref, 1, dev.ptr, host.ctypes.data, bad, None)
    with pytest.raises(ValueError):
        dev.set_valid(np.zeros((4, 6)))


@pytest.mark.gpu
@pytest.mark.parametrize("shape,comp,dtype", CASES)
def test_transfer_windows(shape, comp, dtype):
    pde_hip.get_backend("hip")
    _roundtrip(shape, comp, dtype, np.random.default_rng(3))


@pytest.mark.gpu
def test_transfer_errors():
    _errors(pde_hip.get_backend("hip")._lib)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,comp,dtype", [
    ((130, 140, 261), (), np.float64),     # 38 MB: all copy threads, last chunk partial, rows end inside chunks
    ((96, 100, 130), (3,), np.float32),    # 15 MB: one lane, several chunks
    ((3000, 2100), (), np.float64),        # 50 MB, 2-D
])
def test_transfer_large(shape, comp, dtype):
    pde_hip.get_backend("hip")
    rng = np.random.default_rng(5)
    info = GridInfo(shape, (1.0,) * len(shape), np.dtype(dtype))
    full, view = _window(rng, shape, comp, dtype)
    dev = DeviceArray(info, comp).set_valid(view)
    np.testing.assert_array_equal(dev.get_valid(), view)
    target_full = np.full_like(full, -7)
    target = target_full[(Ellipsis,) + tuple(slice(1, -1) for _ in shape)]
    dev.get_valid(out=target)
    np.testing.assert_array_equal(target, view)
    target[...] = -7
    assert (target_full == -7).all()
    # contiguous upload of the same data through the threaded path
    dev2 = DeviceArray(info, comp).set_valid(np.ascontiguousarray(view))
    np.testing.assert_array_equal(dev2.get_valid(), view)
    # plain pdehip_memcpy_* of a large contiguous buffer (threaded path) round-trips as well
    lib = pde_hip.get_backend("hip")._lib
    flat = rng.random(6_000_001)
    from pde_hip.device import DeviceBuffer

    buf = DeviceBuffer(flat.nbytes)
    lib.memcpy_h2d(buf.ptr, flat.ctypes.data, flat.nbytes, None)
    back = np.empty_like(flat)
    lib.memcpy_d2h(back.ctypes.data, buf.ptr, back.nbytes, None)
    np.testing.assert_array_equal(back, flat)
    buf.free()
