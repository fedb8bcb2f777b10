"""pytest configuration: `gpu` marker, import paths, golden fixtures."""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "py-pde_amd", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


_SHIM = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a HIP device (MI355X); run with -m gpu on the GPU box")
    # development aid: PDEHIP_TEST_SHIM=1 (or =fused) runs the host logic of `-m gpu` tests against the tests-only host
    # shim of the C ABI on a box without a GPU (tests/shim); never set on the GPU box, where the real library must load
    import os

    mode = os.environ.get("PDEHIP_TEST_SHIM", "")
    if mode:
        global _SHIM
        import shimlib

        _SHIM = shimlib.use_shim(fused=mode == "fused")
        _SHIM.__enter__()


def pytest_sessionfinish(session, exitstatus):
    import refpath

    if refpath.REAL:
        refpath.log_loaded_libraries("pytest")


def pytest_unconfigure(config):
    global _SHIM
    if _SHIM is not None:
        _SHIM.__exit__(None, None, None)
        _SHIM = None


@pytest.fixture(scope="session")
def golden_ops():
    return np.load(ROOT / "tests" / "golden" / "ops.npz", allow_pickle=False)


@pytest.fixture(scope="session")
def golden_steppers():
    return np.load(ROOT / "tests" / "golden" / "steppers.npz", allow_pickle=False)


@pytest.fixture()
def rng():
    return np.random.default_rng(0)
