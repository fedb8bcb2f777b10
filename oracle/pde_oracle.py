"""ctypes loader for the CPU oracle (``oracle/pde_oracle.c``).

TEST INFRASTRUCTURE ONLY — the parity checker of the HIP library and the ``cpu_baseline``
leg of ``bench.py``.  May be imported by ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline, never by the product package ``py-pde_amd/pde_hip``.

Parity status: pinned against the reference (see the header of ``pde_oracle.c``).
All arrays are host numpy arrays in the "full" (ghost padded) layout unless noted.
"""

from __future__ import annotations

import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_ROOT = _HERE.parent
if str(_ROOT / "py-pde_amd") not in sys.path:
    sys.path.insert(0, str(_ROOT / "py-pde_amd"))

from pde_hip import _abi  # noqa: E402  (struct definitions only, no device code)

_LIB = None


def build(force: bool = False) -> Path:
    """Compile ``libpde_oracle.so`` with gcc (see ``oracle/Makefile``)."""
    so = _HERE / "libpde_oracle.so"
    if force or not so.exists():
        subprocess.run(["make", "-C", str(_HERE), "-B" if force else "-s"], check=True)
    return so


def lib():
    """Load (building first if necessary) the oracle library and bind its prototypes."""
    global _LIB
    if _LIB is None:
        so = build()
        handle = C.CDLL(str(so))
        for name, (args, _) in _abi.COMPUTE_PROTOTYPES.items():
            if name in {"hostfull_to_full", "full_to_hostfull"}:
                continue  # the oracle's full layout IS the reference's compact host layout
            fn = getattr(handle, "oracle_" + name)
            fn.argtypes = args
            fn.restype = C.c_int
        handle.oracle_set_corner_points_2d.argtypes = [C.POINTER(_abi.Grid), C.POINTER(C.c_int), C.c_void_p]
        handle.oracle_set_corner_points_2d.restype = C.c_int
        # OpenMP threads: what this process may really use (affinity mask and cgroup CPU quota) — the GPU boxes show 256
        # logical CPUs but grant 16: 256 threads on 16 CPUs run the loops tens of times slower
        import os

        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:
            quota, period = (_ROOT / ".." / ".." / "sys" / "fs" / "cgroup" / "cpu.max").resolve().read_text().split()
        except (OSError, ValueError):
            try:
                quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
            except (OSError, ValueError):
                quota, period = "max", "1"
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
        if "OMP_NUM_THREADS" not in os.environ:
            try:
                C.CDLL("libgomp.so.1").omp_set_num_threads(cores)
            except OSError:
                pass
        _LIB = handle
    return _LIB


def _p(arr: np.ndarray | None):
    if arr is None:
        return None
    assert arr.flags.c_contiguous, "oracle needs C-contiguous arrays"
    return arr.ctypes.data_as(C.c_void_p)


def _check(rc: int, name: str):
    if rc != 0:
        msg = f"oracle_{name} failed with code {rc}"
        raise RuntimeError(msg)


def full_shape(shape) -> tuple[int, ...]:
    return tuple(int(s) + 2 for s in shape)


def grid_struct(shape, dx, dtype=np.float64) -> _abi.Grid:
    return _abi.make_grid(shape, dx, dtype)


def valid_to_full(shape, valid: np.ndarray) -> np.ndarray:
    """Embed valid data (…, *shape) into a zero-initialised full array."""
    nd = len(shape)
    lead = valid.shape[: valid.ndim - nd]
    full = np.zeros(lead + full_shape(shape), dtype=valid.dtype)
    full[(...,) + (slice(1, -1),) * nd] = valid
    return full


def set_ghost_cells(g: _abi.Grid, ncomp: int, faces, data_full: np.ndarray) -> None:
    _check(lib().oracle_set_ghost_cells(C.byref(g), ncomp, faces, _p(data_full)), "set_ghost_cells")


def _out_array(g: _abi.Grid, ncomp_lead: tuple[int, ...], layout: int, dtype) -> np.ndarray:
    shape = tuple(g.shape[a] for a in range(g.ndim))
    if layout == _abi.OUT_FULL:
        return np.zeros(ncomp_lead + full_shape(shape), dtype=dtype)
    return np.empty(ncomp_lead + shape, dtype=dtype)


def laplace(g, arr_full, layout=_abi.OUT_VALID):
    out = _out_array(g, (), layout, arr_full.dtype)
    _check(lib().oracle_laplace(C.byref(g), _p(arr_full), _p(out), layout), "laplace")
    return out


def laplace_spectral(g, arr_full, layout=_abi.OUT_VALID):
    """FFT-based Laplacian of periodic 1-D / 2-D grids (pde/backends/numba/operators/cartesian.py:232-330)."""
    out = _out_array(g, (), layout, arr_full.dtype)
    _check(lib().oracle_laplace_spectral(C.byref(g), _p(arr_full), _p(out), layout), "laplace_spectral")
    return out


def set_corner_points_2d(g, periodic, arr_full) -> None:
    per = (C.c_int * 2)(*[int(bool(p)) for p in periodic])
    _check(lib().oracle_set_corner_points_2d(C.byref(g), per, _p(arr_full)), "set_corner_points_2d")


def laplace9(g, periodic, corner_weight, arr_full, layout=_abi.OUT_VALID):
    """Nine-point 2-D Laplacian; writes the corner ghost cells of ``arr_full`` like the reference."""
    out = _out_array(g, (), layout, arr_full.dtype)
    per = (C.c_int * 2)(*[int(bool(p)) for p in periodic])
    _check(lib().oracle_laplace9(C.byref(g), per, float(corner_weight), _p(arr_full), _p(out), layout), "laplace9")
    return out


def gradient(g, arr_full, method="central", layout=_abi.OUT_VALID):
    out = _out_array(g, (g.ndim,), layout, arr_full.dtype)
    _check(lib().oracle_gradient(C.byref(g), _abi.METHODS[method], _p(arr_full), _p(out), layout), "gradient")
    return out


def divergence(g, arr_full, method="central", layout=_abi.OUT_VALID):
    out = _out_array(g, (), layout, arr_full.dtype)
    _check(lib().oracle_divergence(C.byref(g), _abi.METHODS[method], _p(arr_full), _p(out), layout), "divergence")
    return out


def gradient_squared(g, arr_full, central=True, layout=_abi.OUT_VALID):
    out = _out_array(g, (), layout, arr_full.dtype)
    _check(lib().oracle_gradient_squared(C.byref(g), int(central), _p(arr_full), _p(out), layout), "gradient_squared")
    return out


def axis_derivative(g, arr_full, axis, order=1, method="central", layout=_abi.OUT_VALID):
    out = _out_array(g, (), layout, arr_full.dtype)
    _check(lib().oracle_axis_derivative(C.byref(g), axis, order, _abi.METHODS[method], _p(arr_full), _p(out), layout), "axis_derivative")
    return out


def laplace_scaled(g, arr_full, s1, s2):
    out = np.zeros_like(arr_full)
    _check(lib().oracle_laplace_scaled(C.byref(g), _p(arr_full), _p(out), s1, s2), "laplace_scaled")
    return out


def laplace_euler(g, arr_full, y_full, s1, s2):
    out = np.zeros_like(arr_full)
    _check(lib().oracle_laplace_euler(C.byref(g), _p(arr_full), _p(y_full), _p(out), s1, s2), "laplace_euler")
    return out


def cahn_hilliard_mu(g, c_full, gamma):
    out = np.zeros_like(c_full)
    _check(lib().oracle_cahn_hilliard_mu(C.byref(g), _p(c_full), _p(out), gamma), "cahn_hilliard_mu")
    return out


def _ptr_array(arrays):
    arr = (C.c_void_p * len(arrays))()
    for i, a in enumerate(arrays):
        arr[i] = a.ctypes.data
    return arr


def lincomb(g, ncomp, y_full, coefs, ks):
    out = np.zeros_like(ks[0])
    cf = (C.c_double * len(coefs))(*coefs)
    _check(lib().oracle_lincomb(C.byref(g), ncomp, _p(out), _p(y_full), len(ks), cf, _ptr_array(ks)), "lincomb")
    return out


def rk4_combine(g, ncomp, y_full, k1, k2, k3, k4):
    _check(lib().oracle_rk4_combine(C.byref(g), ncomp, _p(y_full), _p(k1), _p(k2), _p(k3), _p(k4)), "rk4_combine")
    return y_full


def ab2_combine(g, ncomp, y_full, rate_cur, rate_prev, dt):
    _check(lib().oracle_ab2_combine(C.byref(g), ncomp, _p(y_full), _p(rate_cur), _p(rate_prev), dt), "ab2_combine")
    return y_full


def adams_bashforth_run(g, rhs, y_full, dt, steps):
    """pde/solvers/adams_bashforth.py:31-70: prev = y - dt*rhs(y); then `steps` AB2 steps (rates are unscaled: dt = 1)."""
    y = y_full
    rate_cur = rhs_scaled(g, rhs, y, 1.0)
    lincomb_out = lincomb(g, 1, y, [-dt], [rate_cur])          # y - dt * rhs(y)
    rate_prev = rhs_scaled(g, rhs, lincomb_out, 1.0)
    for _ in range(steps):
        rate_cur = rhs_scaled(g, rhs, y, 1.0)
        ab2_combine(g, 1, y, rate_cur, rate_prev, dt)
        rate_prev = rate_cur                                   # = rhs(previous state): recomputed by the reference
    return y


def rkf45_combine(g, ncomp, y_full, ks):
    ynew = np.zeros_like(y_full)
    err = C.c_double(0)
    _check(lib().oracle_rkf45_combine(C.byref(g), ncomp, _p(y_full), _p(ynew), _ptr_array(ks), C.addressof(err)), "rkf45_combine")
    return ynew, err.value


def euler_adaptive_combine(g, ncomp, y_full, rate_full, dt, half_full, k_full):
    """End of an adaptive Euler attempt (pde/solvers/euler.py:238-256): returns (step_small, error)."""
    out = np.zeros_like(y_full)
    err = C.c_double(0)
    _check(lib().oracle_euler_adaptive_combine(C.byref(g), ncomp, _p(y_full), _p(rate_full), float(dt), _p(half_full), _p(k_full), _p(out), C.byref(err)),
           "euler_adaptive_combine")
    return out, err.value


def max_abs_pairs(g, npairs, arr_full):
    """max |z| of complex data held as pairs of real components (2p: real part, 2p + 1: imaginary part)."""
    out = C.c_double(0)
    _check(lib().oracle_max_abs_pairs(C.byref(g), npairs, _p(arr_full), C.byref(out)), "max_abs_pairs")
    return out.value


def max_abs_diff(g, ncomp, a_full, b_full):
    err = C.c_double(0)
    _check(lib().oracle_max_abs_diff(C.byref(g), ncomp, _p(a_full), _p(b_full), C.addressof(err)), "max_abs_diff")
    return err.value


def make_rhs(kind, param, bc_c, bc_mu=None, scratch_mu: np.ndarray | None = None) -> _abi.RHS:
    r = _abi.RHS()
    r.kind = kind
    r.param = float(param)
    for i in range(2 * _abi.MAX_DIM):
        r.bc_c[i] = bc_c[i]
        if bc_mu is not None:
            r.bc_mu[i] = bc_mu[i]
    r.scratch_mu = scratch_mu.ctypes.data if scratch_mu is not None else None
    return r


def rhs_scaled(g, rhs, y_full, dt):
    out = np.zeros_like(y_full)
    _check(lib().oracle_rhs_scaled(C.byref(g), C.byref(rhs), _p(y_full), _p(out), dt), "rhs_scaled")
    return out


def euler_run(g, rhs, state_full: np.ndarray, dt: float, nsteps: int) -> np.ndarray:
    """Advance ``state_full`` by ``nsteps`` Euler steps; returns the final full array."""
    a = np.ascontiguousarray(state_full).copy()
    b = np.zeros_like(a)
    res = C.c_void_p()
    _check(lib().oracle_euler_run(C.byref(g), C.byref(rhs), _p(a), _p(b), dt, nsteps, C.byref(res)), "euler_run")
    return a if res.value == a.ctypes.data else b


def rk4_step(g, rhs, y_full, dt):
    work = [np.zeros_like(y_full) for _ in range(5)]
    _check(lib().oracle_rk4_step(C.byref(g), C.byref(rhs), _p(y_full), _ptr_array(work), dt), "rk4_step")
    return y_full


def rkf45_attempt(g, rhs, y_full, dt):
    work = [np.zeros_like(y_full) for _ in range(7)]
    ynew = np.zeros_like(y_full)
    err = C.c_double(0)
    _check(lib().oracle_rkf45_attempt(C.byref(g), C.byref(rhs), _p(y_full), _p(ynew), _ptr_array(work), dt, C.addressof(err)), "rkf45_attempt")
    return ynew, err.value
