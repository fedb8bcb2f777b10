"""Shared helpers of the test-suite (host side; the oracle is used as the CHECKER only)."""

from __future__ import annotations

import json
from pathlib import Path

import numpy as np

import pde_hip
from pde_hip import _abi
from pde_hip.backend import convert_bcs

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"


class HostBuf:
    """Host stand-in for a DeviceBuffer so BC tables can point at numpy memory (oracle side)."""

    def __init__(self, arr: np.ndarray):
        self.arr = np.ascontiguousarray(arr, dtype=np.float64)
        self.ptr = self.arr.ctypes.data


def host_faces(bcs, comp_shape=(), skip=None):
    """BC table whose array pointers are HOST pointers (for the oracle)."""
    return convert_bcs(bcs, comp_shape, skip=skip, upload=HostBuf)


def load_cases(npz) -> list[dict]:
    return json.loads(str(npz["cases"]))


def case_ids(name: str) -> list[str]:
    npz = np.load(GOLDEN / name, allow_pickle=False)
    return [c["id"] for c in load_cases(npz)]


def get_case(npz, cid: str) -> dict:
    return next(c for c in load_cases(npz) if c["id"] == cid)


def make_grid(case: dict) -> pde_hip.CartesianGrid:
    return pde_hip.CartesianGrid(case["bounds"], case["shape"], periodic=case["periodic"])


def oracle_grid(grid, dtype=np.float64) -> _abi.Grid:
    return _abi.make_grid(grid.shape, grid.discretization, dtype)


def to_full(grid, valid: np.ndarray) -> np.ndarray:
    """Embed valid data into a zero-initialised compact full array (reference host layout)."""
    nd = grid.num_axes
    lead = valid.shape[: valid.ndim - nd]
    full = np.zeros(lead + grid._shape_full, dtype=valid.dtype)
    full[(...,) + (slice(1, -1),) * nd] = valid
    return full


def interior(grid, full: np.ndarray) -> np.ndarray:
    return full[(...,) + (slice(1, -1),) * grid.num_axes]


def face_mask(grid, lead_shape=()) -> np.ndarray:
    """Boolean mask of the cells the reference defines: interior + face ghosts (no edges/corners)."""
    shape = grid._shape_full
    nd = len(shape)
    idx = np.indices(shape)
    n_ghost = sum(((idx[a] == 0) | (idx[a] == shape[a] - 1)).astype(int) for a in range(nd))
    return np.broadcast_to(n_ghost <= 1, tuple(lead_shape) + shape)


def max_rel(a: np.ndarray, b: np.ndarray) -> float:
    """The parity metric of BASELINE.md §3: max|a-b| / max|b|."""
    denom = np.abs(b).max()
    return float(np.abs(a - b).max() / (denom if denom > 0 else 1.0))
