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


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a HIP device (MI355X); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def golden_ops():
    return np.load(ROOT / "tests" / "golden" / "ops.npz", allow_pickle=False)


@pytest.fixture(scope="session")
def golden_steppers():
    return np.load(ROOT / "tests" / "golden" / "steppers.npz", allow_pickle=False)


@pytest.fixture()
def rng():
    return np.random.default_rng(0)
