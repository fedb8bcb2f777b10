"""ctypes mirror of ``include/pdehip.h`` (structs, enums and prototype table).

The same table is used to bind ``libpdehip.so`` (the product) and — by ``oracle/pde_oracle.py``
only, for tests — the CPU oracle, whose entry points have the same signatures without the
trailing stream argument.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

MAX_DIM = 3
ABI_VERSION = 7

F64, F32 = 0, 1
CENTRAL, FORWARD, BACKWARD = 0, 1, 2
OUT_VALID, OUT_FULL = 0, 1
BC_SKIP, BC_ORDER1, BC_ORDER2 = 0, 1, 2
BCF_ARRAYS, BCF_NORMAL = 1, 2
RHS_DIFFUSION, RHS_CAHN_HILLIARD = 0, 1

METHODS = {"central": CENTRAL, "forward": FORWARD, "backward": BACKWARD}

_DTYPES = {np.dtype(np.float64): F64, np.dtype(np.float32): F32}


def dtype_code(dtype) -> int:
    """Translate a numpy dtype to the PDEHIP_F* code; complex/ints are not supported."""
    dt = np.dtype(dtype)
    try:
        return _DTYPES[dt]
    except KeyError:
        msg = f"hip backend supports float64 and float32 fields only (got {dt})"
        raise NotImplementedError(msg) from None


class Grid(C.Structure):
    _fields_ = [
        ("ndim", C.c_int32),
        ("dtype", C.c_int32),
        ("shape", C.c_int64 * MAX_DIM),
        ("dx", C.c_double * MAX_DIM),
    ]


class BCFace(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("flags", C.c_int32),
        ("index1", C.c_int64),
        ("index2", C.c_int64),
        ("const_v", C.c_double),
        ("factor1", C.c_double),
        ("factor2", C.c_double),
        ("const_arr", C.c_void_p),
        ("factor1_arr", C.c_void_p),
        ("factor2_arr", C.c_void_p),
    ]


FaceArray = BCFace * (2 * MAX_DIM)


class RHS(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("reserved", C.c_int32),
        ("param", C.c_double),
        ("bc_c", FaceArray),
        ("bc_mu", FaceArray),
        ("scratch_mu", C.c_void_p),
        ("bc_program", C.c_void_p),
        ("t", C.c_double),
    ]


class BcProgFace(C.Structure):
    """``pdehip_bcprog_face_t``: where one expression face writes its coefficient arrays and how its cells map to coordinates."""

    _fields_ = [("const_arr", C.c_void_p), ("factor_arr", C.c_void_p), ("m1", C.c_int64), ("m2", C.c_int64),
                ("origin", C.c_double * 3), ("step", C.c_double * 3), ("index", C.c_int32 * 3), ("reads_value", C.c_int32), ("dx", C.c_double),
                ("axis", C.c_int32), ("component", C.c_int32), ("value_index", C.c_int64), ("first", C.c_int64 * 3)]


JIT_NONE = -(2**31)   # PDEHIP_JIT_NONE


class Exchange(C.Structure):
    """``pdehip_exchange_t``: where the ghost layers of a pass's operand come from on a decomposed grid."""

    _fields_ = [("comm", C.c_void_p), ("blocks", C.c_int32), ("lower", C.c_int32), ("upper", C.c_int32), ("nb6", C.c_int32 * 6)]


class JitPass(C.Structure):
    """``pdehip_jit_pass_t``: one pass of the Euler loop of an expression PDE (``pdehip_jit_euler_run``)."""

    _fields_ = [("handle", C.c_void_p), ("src", C.c_int32), ("extras", C.c_int32 * 3), ("out", C.c_int32), ("faces", C.c_void_p),
                ("exchange", C.c_void_p)]


class Adaptive(C.Structure):
    """``pdehip_adaptive_t``: controller state of the adaptive loop (``pdehip_slab_rkf45_run``)."""

    _fields_ = [
        ("t_start", C.c_double),
        ("t_end", C.c_double),
        ("dt", C.c_double),
        ("tolerance", C.c_double),
        ("dt_min", C.c_double),
        ("dt_max", C.c_double),
        ("t_last", C.c_double),
        ("steps", C.c_int64),
        ("attempts", C.c_int64),
        ("stat_count", C.c_int64),
        ("stat_min", C.c_double),
        ("stat_max", C.c_double),
        ("stat_mean", C.c_double),
        ("stat_m2", C.c_double),
    ]


def adaptive_statistics(ctl: Adaptive) -> dict:
    """Statistics of the accepted step sizes in the format of ``OnlineStatistics.to_dict`` (pde/tools/math.py:125-174)."""
    import math

    n = int(ctl.stat_count)
    var = ctl.stat_m2 / (n - 1) if n >= 2 else math.nan
    return {"min": ctl.stat_min if n else math.inf, "max": ctl.stat_max if n else -math.inf, "mean": ctl.stat_mean,
            "std": math.sqrt(var) if var == var else math.nan, "count": n}


def make_grid(shape, dx, dtype) -> Grid:
    """Build the POD grid descriptor from ``grid.shape`` / ``grid.discretization``."""
    shape = tuple(int(s) for s in shape)
    if not 1 <= len(shape) <= MAX_DIM:
        msg = f"hip backend supports 1 to {MAX_DIM} dimensional Cartesian grids"
        raise NotImplementedError(msg)
    g = Grid()
    g.ndim = len(shape)
    g.dtype = dtype_code(dtype)
    for a in range(MAX_DIM):
        g.shape[a] = shape[a] if a < len(shape) else 1
        g.dx[a] = float(dx[a]) if a < len(shape) else 1.0
    return g


_vp = C.c_void_p
_pg = C.POINTER(Grid)
_pf = C.POINTER(BCFace)
_pr = C.POINTER(RHS)
_pd = C.POINTER(C.c_double)
_pvp = C.POINTER(C.c_void_p)
_pa = C.POINTER(Adaptive)
_i, _i64, _d = C.c_int, C.c_int64, C.c_double

# name -> argument types WITHOUT the trailing stream argument.  ``True`` in the second slot
# marks functions that take a stream in libpdehip (the oracle twin drops it).
COMPUTE_PROTOTYPES: dict[str, tuple[list, bool]] = {
    "valid_to_full": ([_pg, _i, _vp, _vp], True),
    "full_to_valid": ([_pg, _i, _vp, _vp], True),
    "hostfull_to_full": ([_pg, _i, _vp, _vp], True),
    "full_to_hostfull": ([_pg, _i, _vp, _vp], True),
    "set_ghost_cells": ([_pg, _i, _pf, _vp], True),
    "laplace": ([_pg, _vp, _vp, _i], True),
    "laplace_spectral": ([_pg, _vp, _vp, _i], True),
    "gradient": ([_pg, _i, _vp, _vp, _i], True),
    "divergence": ([_pg, _i, _vp, _vp, _i], True),
    "gradient_squared": ([_pg, _i, _vp, _vp, _i], True),
    "axis_derivative": ([_pg, _i, _i, _i, _vp, _vp, _i], True),
    "laplace9": ([_pg, C.POINTER(_i), _d, _vp, _vp, _i], True),
    "laplace_scaled": ([_pg, _vp, _vp, _d, _d], True),
    "laplace_euler": ([_pg, _vp, _vp, _vp, _d, _d], True),
    "cahn_hilliard_mu": ([_pg, _vp, _vp, _d], True),
    "lincomb": ([_pg, _i, _vp, _vp, _i, _pd, _pvp], True),
    "rk4_combine": ([_pg, _i, _vp, _vp, _vp, _vp, _vp], True),
    "rkf45_combine": ([_pg, _i, _vp, _vp, _pvp, _vp], True),
    "ab2_combine": ([_pg, _i, _vp, _vp, _vp, _d], True),
    "euler_adaptive_combine": ([_pg, _i, _vp, _vp, _d, _vp, _vp, _vp, _vp], True),
    "max_abs_diff": ([_pg, _i, _vp, _vp, _vp], True),
    "max_abs_pairs": ([_pg, _i, _vp, _vp], True),
    "integrate": ([_pg, _i, _vp, _d, _vp], True),
    "count_nonfinite": ([_pg, _i, _vp, _vp], True),
    "add_gaussian_noise": ([_pg, _i, _vp, _d, C.c_uint64, C.c_uint64, C.c_uint64], True),
    "rhs_scaled": ([_pg, _pr, _vp, _vp, _d], True),
    "euler_run": ([_pg, _pr, _vp, _vp, _d, _i64, _pvp], True),
    "rk4_step": ([_pg, _pr, _vp, _pvp, _d], True),
    "rkf45_attempt": ([_pg, _pr, _vp, _vp, _pvp, _d, _vp], True),
}

RUNTIME_PROTOTYPES: dict[str, tuple[list, object]] = {
    "last_error": ([], C.c_char_p),
    "last_kernel_name": ([], C.c_char_p),
    "set_fastmath": ([_i], _i),
    "get_fastmath": ([C.POINTER(_i)], _i),
    "abi_version": ([], _i),
    "device_count": ([C.POINTER(_i)], _i),
    "set_device": ([_i], _i),
    "device_name": ([C.c_char_p, C.c_size_t], _i),
    "layout": ([_pg, C.POINTER(C.c_int64)], _i),
    "malloc": ([_pvp, C.c_size_t], _i),
    "free": ([_vp], _i),
    "memset": ([_vp, _i, C.c_size_t, _vp], _i),
    "memcpy_h2d": ([_vp, _vp, C.c_size_t, _vp], _i),
    "memcpy_d2h": ([_vp, _vp, C.c_size_t, _vp], _i),
    "memcpy_d2d": ([_vp, _vp, C.c_size_t, _vp], _i),
    "copy_nt": ([_vp, _vp, C.c_size_t, _vp], _i),
    "upload_valid": ([_pg, _i, _vp, C.POINTER(C.c_int64), _vp, _vp], _i),
    "download_valid": ([_pg, _i, _vp, _vp, C.POINTER(C.c_int64), _vp], _i),
    "stream_create": ([_pvp], _i),
    "stream_destroy": ([_vp], _i),
    "stream_synchronize": ([_vp], _i),
    "stream_wait_event": ([_vp, _vp], _i),
    "event_create": ([_pvp], _i),
    "event_destroy": ([_vp], _i),
    "event_record": ([_vp, _vp], _i),
    "event_synchronize": ([_vp], _i),
    "event_elapsed_ms": ([_vp, _vp, C.POINTER(C.c_float)], _i),
}


# slab-parallel layer (pdehip_comm.hip): name -> full argument list
COMM_PROTOTYPES: dict[str, list] = {
    "comm_unique_id": [C.c_char_p, _vp],
    "comm_create": [C.c_char_p, _vp, _i, _i, _pvp],
    "comm_destroy": [_vp],
    "halo_exchange": [_vp, _pg, _vp, _i, _i, _vp],
    "allreduce_max": [_vp, _vp, _vp],
    "slab_euler_run": [_vp, _pg, _pr, _i, _i, _vp, _vp, _d, _i64, _pvp, _vp],
    "comm_info": [_vp, C.POINTER(_i), C.c_char_p, C.c_size_t],
    "slab_euler2_supported": [_pg, _pr, C.POINTER(_i)],
    "slab_euler2_run": [_vp, _pg, _pr, _i, _i, _vp, _vp, _d, _i64, _pvp, _vp],
    "release_scratch": [],
    "slab_euler4_supported": [_pg, _pr, C.POINTER(_i)],
    "slab_euler4_run": [_vp, _pg, _pr, _i, _i, _vp, _vp, _d, _i64, _pvp, _vp],
    "slab_ch_supported": [_pg, _pr, C.POINTER(_i)],
    "slab_ch_sweep": [_vp, _pg, _pr, _i, _i, _vp, _vp, _d, _i, _vp],
    # slab-parallel Runge-Kutta / adaptive loop / generic right-hand side (one C call per run)
    "slab_flags_supported": [_pg, _pr, _i, _i, C.POINTER(_i)],
    "slab_rhs_scaled": [_vp, _pg, _pr, _i, _i, _i, _vp, _vp, _d, _vp],
    "slab_euler_sweeps": [_vp, _pg, _pr, _i, _i, _i, _vp, _vp, _d, _i64, _pvp, _vp],
    "slab_rk4_run": [_vp, _pg, _pr, _i, _i, _i, _vp, _pvp, _d, _i64, _vp],
    "slab_rkf45_run": [_vp, _pg, _pr, _i, _i, _i, _vp, _vp, _pvp, _vp, _pa, _pvp, _vp],
    # the reference's adaptive Euler loop (rate carried from attempt to attempt): slab / serial, one C call per run
    "slab_euler_adaptive_run": [_vp, _pg, _pr, _i, _i, _i, _vp, _vp, _pvp, _vp, _pa, _pvp, _vp],
    "euler_adaptive_run": [_pg, _pr, _vp, _vp, _pvp, _vp, _pa, _pvp, _vp],
    # block decomposition (csrc/pdehip_block_loops.h)
    "block_exchange": [_vp, _pg, C.POINTER(_i), _vp, _vp],
    "block_run": [_vp, _pg, _pr, C.POINTER(_i), _i, _i, _vp, _vp, _pvp, _vp, _d, _i64, _pa, _pvp, _vp],
    "field_product": [_pg, _i, _i, _i, _vp, _vp, _vp, _vp],
    "block2_supported": [_pg, _pr, C.POINTER(_i), C.POINTER(_i)],
    "block2_euler_run": [_vp, _pg, _pr, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), _vp, _vp, _d, _i64, _pvp, _vp],
    # Adams-Bashforth step in one sweep (device only: the oracle runs rhs_scaled + ab2_combine)
    "ab2_step": [_pg, _pr, _vp, _vp, _vp, _vp, _d, C.POINTER(_i), _vp],
    # fixed-step RK4 loop (device only: the oracle loops over rk4_step)
    "rk4_run": [_pg, _pr, _vp, _pvp, _d, _i64, _vp],
    # two Euler steps per sweep (device only: the oracle takes two single steps)
    "diffusion_euler2": [_pg, _pf, _vp, _vp, _d, _d, C.POINTER(_i), _vp],
    "euler_multi_2d": [_pg, _pr, _vp, _vp, _d, _i, C.POINTER(_i), _vp],
    "diffusion_euler2_slab": [_pg, _pf, _vp, _vp, _d, _d, _i, C.POINTER(_i), _vp],
    "cahn_hilliard_fused": [_pg, _pf, _pf, _vp, _vp, _d, _d, _i, C.POINTER(_i), _vp],
    # run-time specialised expression kernels (pdehip_jit.hip)
    "jit_create": [C.c_char_p, _pvp],
    "jit_destroy": [_vp],
    "jit_check": [_vp, _i, _i],
    "jit_apply": [_vp, _pg, _vp, _pvp, _vp, _pd, _i, _pf, _vp],
    "jit_apply_stage": [_vp, _pg, _vp, _pvp, _vp, _pd, _i, _pf, _i, _vp, _i, _pvp, _pd, _d, _vp, _vp, C.POINTER(_i), _vp],
    "jit_euler2": [_vp, _pg, _vp, _vp, _pd, _i, _pf, C.POINTER(_i), _vp],
    "jit_create2": [C.c_char_p, C.c_char_p, _pvp],
    "jit_fused2": [_vp, _pg, _vp, _vp, _pd, _i, _pf, _pf, C.POINTER(_i), _vp],
    "jit_euler_run": [_pg, C.POINTER(JitPass), _i, _pvp, _i, _vp, _vp, _i, _d, _d, _i, _i64, _vp, _pvp, _vp],
    "jit_rk_run": [_pg, C.POINTER(JitPass), _i, _pvp, _i, _i, _vp, _vp, _pvp, _vp, _d, _d, _i64, _pa, _i, _vp, _pvp, _vp],
    "jit_euler_adaptive_run": [_pg, C.POINTER(JitPass), _i, _pvp, _i, _i, _vp, _vp, _pvp, _vp, _pa, _i, _vp, _pvp, _vp],
    # expression boundary conditions evaluated on the device
    "bcprog_create": [C.c_char_p, _i, C.POINTER(BcProgFace), _pg, _pvp],
    "bcprog_run": [_vp, _d, _vp, _vp],
    "bcprog_destroy": [_vp],
}


def exported_symbols() -> list[str]:
    """All symbols ``include/pdehip.h`` declares (checked by tests/test_cabi.py)."""
    return ["pdehip_" + n for n in list(RUNTIME_PROTOTYPES) + list(COMPUTE_PROTOTYPES) + list(COMM_PROTOTYPES)]
