"""CPU checks of the C-ABI boundary: the library loads, exports every declared symbol, and the
ctypes mirror agrees with include/pdehip.h.  No compute calls (there is no GPU here)."""

from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from pde_hip import _abi, _lib

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "pdehip.h").read_text()


def declared_symbols() -> set[str]:
    return set(re.findall(r"\b(pdehip_[a-z0-9_]+)\s*\(", HEADER))


def test_header_and_ctypes_table_agree():
    assert declared_symbols() == set(_abi.exported_symbols())


def test_every_declaration_cites_the_reference():
    """Each compute entry point documents the reference interface it replaces (file:line)."""
    assert len(re.findall(r"pde/[a-z_/]+\.py:\d+", HEADER)) >= 25


@pytest.fixture(scope="module")
def handle():
    if not _lib.LIB_PATH.exists():
        pytest.fail(f"{_lib.LIB_PATH} missing - run __graft_entry__.build() first")
    return C.CDLL(str(_lib.LIB_PATH))


def test_library_exports_all_symbols(handle):
    for name in sorted(declared_symbols()):
        assert hasattr(handle, name), f"libpdehip.so does not export {name}"


def test_abi_version_and_error_string(handle):
    handle.pdehip_abi_version.restype = C.c_int
    assert handle.pdehip_abi_version() == _abi.ABI_VERSION
    handle.pdehip_last_error.restype = C.c_char_p
    assert isinstance(handle.pdehip_last_error(), bytes)


def test_struct_layout_matches_header():
    """Field order/size of the POD structs (catches drift between header and ctypes)."""
    assert C.sizeof(_abi.Grid) == 4 + 4 + 3 * 8 + 3 * 8
    assert C.sizeof(_abi.BCFace) == 4 + 4 + 2 * 8 + 3 * 8 + 3 * 8
    assert C.sizeof(_abi.RHS) == 4 + 4 + 8 + 2 * 6 * C.sizeof(_abi.BCFace) + 8 + 8 + 8    # ... scratch_mu, bc_program, t (ABI version 2)
    assert C.sizeof(_abi.BcProgFace) == 2 * 8 + 2 * 8 + 2 * 3 * 8 + 3 * 4 + 4 + 8 + 4 + 4 + 8 + 3 * 8    # ... axis, component, value_index, first[3] (ABI version 4)
    for struct, cname in [(_abi.Grid, "pdehip_grid"), (_abi.BCFace, "pdehip_bc_face"), (_abi.RHS, "pdehip_rhs"), (_abi.Adaptive, "pdehip_adaptive"),
                          (_abi.BcProgFace, "pdehip_bcprog_face")]:
        body = re.search(r"typedef struct " + cname + r" \{(.*?)\} " + cname + "_t;", HEADER, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = re.findall(r"\**\s*([a-z_0-9]+)(?:\[[^\]]*\])?\s*(?:,|;)", body)
        assert [n for n in names if n] == [f[0] for f in struct._fields_], cname


def test_host_side_fails_loudly_without_gpu():
    """No silent CPU fallback: without a HIP device every compute entry point refuses (constructing the backend object
    alone must work — py-pde instantiates all registered backends just to list operators)."""
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    import pde_hip

    backend = pde_hip.get_backend("hip")
    with pytest.raises(RuntimeError, match="no HIP device"):
        backend.device_name
    with pytest.raises(RuntimeError, match="no HIP device"):
        backend.numpy_to_native(np.zeros((4, 4)), grid=pde_hip.UnitGrid([4, 4]))
    grid = pde_hip.UnitGrid([4, 4])
    with pytest.raises(RuntimeError, match="no HIP device"):
        pde_hip.ScalarField(grid, 1.0).laplace("auto_periodic_neumann")
    with pytest.raises(RuntimeError, match="no HIP device"):
        pde_hip.DiffusionPDE().solve(pde_hip.ScalarField(grid, 1.0), 1.0, dt=0.1)


def test_the_library_cannot_be_redirected_by_one_environment_variable():
    """PDEHIP_LIB alone must not swap the product's library: without PDEHIP_ALLOW_LIB_OVERRIDE=1 the import fails."""
    import os
    import subprocess
    import sys

    code = "import sys; sys.path[:0] = [%r]; from pde_hip import _lib; print(_lib.LIB_PATH)" % str(ROOT / "py-pde_amd")
    env = {k: v for k, v in os.environ.items() if k not in ("PDEHIP_LIB", "PDEHIP_ALLOW_LIB_OVERRIDE")}
    plain = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True)
    assert plain.stdout.strip().endswith("py-pde_amd/lib/libpdehip.so")
    lone = subprocess.run([sys.executable, "-c", code], env={**env, "PDEHIP_LIB": "/tmp/other.so"}, capture_output=True, text=True, check=False)
    assert lone.returncode != 0 and "PDEHIP_ALLOW_LIB_OVERRIDE" in lone.stderr
    both = subprocess.run([sys.executable, "-c", code], env={**env, "PDEHIP_LIB": "/tmp/other.so", "PDEHIP_ALLOW_LIB_OVERRIDE": "1"},
                          capture_output=True, text=True, check=True)
    assert both.stdout.strip() == "/tmp/other.so"
