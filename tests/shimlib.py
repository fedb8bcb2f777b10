"""Host stand-in of libpdehip.so for CPU tests of the Python host side (TESTS ONLY).

``tests/shim/pdehip_shim.c`` implements the C ABI of ``include/pdehip.h`` with host memory and the CPU
oracle as kernels.  ``use_shim()`` builds it and installs it as the library object of ``pde_hip._lib`` for the
duration of a ``with`` block / fixture, so that the plugin class, the BC conversion, the stepper loops and the
``solver.info`` bookkeeping run end to end through the REAL py-pde in a container without a GPU.  Nothing in
``py-pde_amd/`` knows about this file; the product resolves ``py-pde_amd/lib/libpdehip.so`` and needs a HIP device.
"""

from __future__ import annotations

import contextlib
import os
import subprocess
from pathlib import Path

SHIM_DIR = Path(__file__).resolve().parent / "shim"
SHIM_SO = SHIM_DIR / "_build" / "libpdehip_shim.so"
_CFLAGS = ["-O2", "-mavx2", "-fPIC", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-Wall", "-Wextra",
           "-Wno-unused-function", "-Wno-unused-parameter"]


def build(force: bool = False) -> Path:
    """Compile the shim with gcc (seconds); rebuilt when any of its sources is newer."""
    root = SHIM_DIR.parent.parent
    sources = [SHIM_DIR / "pdehip_shim.c", *sorted(SHIM_DIR.glob("*.inc")), *sorted(SHIM_DIR.glob("*.cpp")), *sorted(SHIM_DIR.glob("*.h")),
               root / "oracle" / "pde_oracle.c", root / "oracle" / "pde_oracle_impl.inc", root / "include" / "pdehip.h",
               *sorted((root / "py-pde_amd" / "csrc").glob("pdehip_*loops.h"))]
    if not force and SHIM_SO.exists() and all(SHIM_SO.stat().st_mtime >= s.stat().st_mtime for s in sources if s.exists()):
        return SHIM_SO
    SHIM_SO.parent.mkdir(exist_ok=True)
    objs = []
    cmd = ["gcc", *_CFLAGS, "-c", str(SHIM_DIR / "pdehip_shim.c"), "-o", str(SHIM_SO.parent / "shim.o")]
    comm = SHIM_DIR / "pdehip_shim_comm.cpp"
    if comm.exists():
        cmd.insert(1, "-DSHIM_WITH_COMM")
    subprocess.run(cmd, check=True)
    objs.append(str(SHIM_SO.parent / "shim.o"))
    if comm.exists():
        subprocess.run(["g++", "-O2", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-c", str(comm), "-o", str(SHIM_SO.parent / "comm.o")],
                       check=True)
        objs.append(str(SHIM_SO.parent / "comm.o"))
    tmp = SHIM_SO.with_suffix(f".{os.getpid()}.tmp")
    subprocess.run(["g++", "-shared", "-fopenmp", "-o", str(tmp), *objs, "-lm", "-ldl", "-lpthread"], check=True)
    os.replace(tmp, SHIM_SO)
    return SHIM_SO


def _forget_layouts() -> None:
    """Backend objects cache the array layout of a grid (pitches, component stride) as the library reports it; the shim's
    compact host layout differs from the device's padded one, so the caches are dropped whenever the library is swapped
    (a test that evaluates the same grid with both libraries in one process would otherwise mix them)."""
    import sys

    mod = sys.modules.get("pde_hip.backend")
    if mod is not None:
        for backend in list(getattr(mod, "_BACKENDS", {}).values()):
            backend._info_cache.clear()
    pde_mod = sys.modules.get("pde.backends")
    if pde_mod is not None:
        try:
            hip = pde_mod.backend_registry._backends.get("hip")
        except AttributeError:
            hip = None
        if hip is not None and hasattr(hip, "_info_cache"):
            hip._info_cache.clear()


@contextlib.contextmanager
def use_shim(fused: bool = False, devices: int = 1):
    """Install the shim as ``pde_hip._lib``'s library; restores the previous state on exit."""
    from pde_hip import _lib

    from refpath import REAL

    if REAL:
        # PDEHIP_DROPIN_REAL=1 (tools/gpu_dropin_real.sh, GPU box only): the drop-in tests drive the REAL libpdehip.so
        yield _lib.get_lib()
        return
    so = build()
    saved = (_lib._LIB, _lib._DEVICE, set(_lib._threads_ready))
    env = {k: os.environ.get(k) for k in ("PDEHIP_SHIM_FUSED", "PDEHIP_SHIM_DEVICES")}
    os.environ["PDEHIP_SHIM_FUSED"] = "1" if fused else "0"
    os.environ["PDEHIP_SHIM_DEVICES"] = str(devices)
    _lib._LIB, _lib._DEVICE = _lib._Lib(so), None
    _lib._threads_ready.clear()
    _forget_layouts()
    try:
        yield _lib._LIB
    finally:
        _forget_layouts()
        _lib._LIB, _lib._DEVICE = saved[0], saved[1]
        _lib._threads_ready.clear()
        _lib._threads_ready.update(saved[2])
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
