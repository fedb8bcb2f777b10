"""ctypes binding of ``libpdehip.so`` (the C ABI declared in ``include/pdehip.h``).

The library is built in-tree by ``py-pde_amd/Makefile`` (``__graft_entry__.build()``) into
``py-pde_amd/lib/libpdehip.so``.  There is NO fallback: if the shared object is missing or no
HIP device is visible, every compute entry point raises.
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path

from . import _abi

# The library is the in-tree build.  PDEHIP_LIB=<path> (another build of the same library: A/B timing of kernel variants, the test
# harness's multi-process workers) is honoured ONLY together with PDEHIP_ALLOW_LIB_OVERRIDE=1 - a lone environment variable must not be
# able to redirect the product to some other ABI-compatible object; set without the permission it is an error.
_IN_TREE = Path(__file__).resolve().parent.parent / "lib" / "libpdehip.so"
_OVERRIDE = os.environ.get("PDEHIP_LIB")
if _OVERRIDE and os.environ.get("PDEHIP_ALLOW_LIB_OVERRIDE") != "1":
    msg = ("PDEHIP_LIB is set but PDEHIP_ALLOW_LIB_OVERRIDE=1 is not: refusing to load another library in place of "
           f"{_IN_TREE} (development / test harness only)")
    raise ImportError(msg)
LIB_PATH = Path(_OVERRIDE) if _OVERRIDE else _IN_TREE

_E_VALUE, _E_NOTIMPL = 1, 2


class HipRuntimeError(RuntimeError):
    """An HIP runtime call inside libpdehip failed."""


def _preload_torch_hip_runtime() -> None:
    """Make sure ONE HIP runtime serves the process when PyTorch-ROCm is installed.

    torch wheels bundle their own ``libamdhip64.so`` (same SONAME as the system ROCm one).  The
    multi-GPU layer uses torch for process groups, so both libpdehip and torch live in one process;
    whichever HIP runtime is loaded first is shared by both.  torch only works with its own copy
    (its bundled HSA runtime must match), so that copy is loaded first — without importing torch.
    Set ``PDEHIP_PRELOAD_TORCH_HIP=0`` to use the system runtime (then never import torch afterwards).
    """
    if os.environ.get("PDEHIP_PRELOAD_TORCH_HIP", "1") == "0":
        return
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    cand = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if cand.exists():
        try:
            C.CDLL(str(cand), mode=getattr(os, "RTLD_NOW", 2) | getattr(os, "RTLD_GLOBAL", 0x100))
        except OSError:
            pass


class _Lib:
    """Thin wrapper turning the int status convention into Python exceptions.

    Error mapping: 1 → ValueError, 2 → NotImplementedError,
    anything else → RuntimeError, always with ``pdehip_last_error()`` as message.
    """

    def __init__(self, path: Path):
        if not path.exists():
            msg = (
                f"{path} not found - build it with `make -C {path.parent.parent}` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "The hip backend has no CPU fallback."
            )
            raise ImportError(msg)
        self.path = path
        _preload_torch_hip_runtime()
        self._h = C.CDLL(str(path), mode=getattr(os, "RTLD_NOW", 2))
        for name, (args, res) in _abi.RUNTIME_PROTOTYPES.items():
            fn = getattr(self._h, "pdehip_" + name)
            fn.argtypes = args
            fn.restype = res
        for name, (args, has_stream) in _abi.COMPUTE_PROTOTYPES.items():
            fn = getattr(self._h, "pdehip_" + name)
            fn.argtypes = [*args, C.c_void_p] if has_stream else list(args)
            fn.restype = C.c_int
        for name, args in _abi.COMM_PROTOTYPES.items():
            fn = getattr(self._h, "pdehip_" + name)
            fn.argtypes = list(args)
            fn.restype = C.c_int
        if self._h.pdehip_abi_version() != _abi.ABI_VERSION:
            msg = "libpdehip.so ABI version mismatch - rebuild the library"
            raise ImportError(msg)

    def last_error(self) -> str:
        return self._h.pdehip_last_error().decode(errors="replace")

    def check(self, rc: int) -> None:
        if rc == 0:
            return
        msg = self.last_error()
        if rc == _E_VALUE:
            raise ValueError(msg)
        if rc == _E_NOTIMPL:
            raise NotImplementedError(msg)
        raise HipRuntimeError(f"{msg} [code {rc}]")

    def __getattr__(self, name: str):
        fn = getattr(self._h, "pdehip_" + name)
        if fn.restype is not C.c_int:
            return fn

        def call(*args):
            self.check(fn(*args))

        call.__name__ = name
        setattr(self, name, call)
        return call


_LIB: _Lib | None = None
_DEVICE: int | None = None   # the HIP device this process drives (one process per GPU)


def get_lib() -> _Lib:
    """Load libpdehip.so (once)."""
    global _LIB
    if _LIB is None:
        _LIB = _Lib(LIB_PATH)
    return _LIB


def device_count() -> int:
    n = C.c_int(0)
    try:
        get_lib().device_count(C.byref(n))
    except HipRuntimeError:
        return 0
    return n.value


def default_device() -> int:
    """Device of a process that did not name one: its ``LOCAL_RANK`` under a one-process-per-GPU launcher
    (``torch.distributed.run`` / ``torchrun`` export it), else device 0."""
    try:
        return max(0, int(os.environ.get("LOCAL_RANK", "0")))
    except ValueError:
        return 0


def current_device() -> int | None:
    """Index of the device selected by the first `require_device` call (None before it)."""
    return _DEVICE


def require_device(device: int | None = None) -> _Lib:
    """Return the library after making sure a HIP device is usable; raise loudly otherwise.

    The first call selects the device (``device`` or 0) for the whole process.  Buffers, grids and kernels all
    run on that one device (HIP tracks the current device per host thread, so it is re-applied when another
    thread calls in); asking for a different index later is refused instead of silently mixing devices:
    multi-GPU runs use one process per GPU (``pde_hip/distributed.py``).
    """
    global _DEVICE
    lib = get_lib()
    if _DEVICE is None:
        if device_count() < 1:
            msg = "hip backend: no HIP device visible (MI355X required; there is no CPU fallback)"
            raise RuntimeError(msg)
        dev = default_device() if device is None else int(device)
        lib.set_device(dev)
        _DEVICE = dev
        _threads_ready.add(threading.get_ident())
        if os.environ.get("PDEHIP_FASTMATH", "0") == "1":
            lib.set_fastmath(1)   # opt-in FMA contraction for code that drives the library without a backend object (pde_hip/backend.py: `fastmath`)
    else:
        if device is not None and int(device) != _DEVICE:
            msg = (
                f"hip backend: this process already drives HIP device {_DEVICE}; cannot switch to device {int(device)} "
                "(start one process per GPU)"
            )
            raise RuntimeError(msg)
        ident = threading.get_ident()
        if ident not in _threads_ready:
            lib.set_device(_DEVICE)   # the current device is a per-thread setting of the HIP runtime
            _threads_ready.add(ident)
    return lib


_threads_ready: set[int] = set()
