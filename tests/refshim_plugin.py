"""pytest plugin that runs the REFERENCE's own generic tests against the hip backend (tests only).

Loaded with ``-p refshim_plugin`` by ``tests/test_reference_suite.py`` in a child pytest process that collects
test files straight from ``/root/reference/tests`` (nothing is copied).  It stands in for the reference's
``tests/conftest.py`` (which cannot be imported here: it needs numba and matplotlib; ``--confcutdir`` keeps it
out) and

* installs the host shim (``tests/shim``: the C ABI of include/pdehip.h on host memory + oracle kernels) and
  registers the hip backend with the real py-pde (``pde_hip.pypde_plugin``),
* puts ``"hip"`` into the backend lists the reference parametrises its generic tests with
  (``ALL_BACKENDS`` & co.) — only ``"hip"`` is kept, the other backends are the
  reference's own business,
* provides the ``backend`` / ``rng`` fixtures and the floating-point error policy of the reference's conftest.
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent
for p in (HERE, HERE.parent / "py-pde_amd", HERE.parent):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))
from refpath import add_to_path  # noqa: E402

add_to_path()

import shimlib  # noqa: E402

_shim_ctx = None
BACKEND_LIST_NAMES = ("ALL_BACKENDS", "ALL_COMPILED_BACKENDS", "ALL_BACKENDS_NO_NUMBA")


def pytest_configure(config):
    global _shim_ctx
    import os

    for name in ("interactive", "multiprocessing", "slow"):
        config.addinivalue_line("markers", f"{name}: marker of the reference test-suite")
    _shim_ctx = shimlib.use_shim(fused=os.environ.get("REFSHIM_FUSED", "0") == "1")
    _shim_ctx.__enter__()
    import pde
    import pde_hip.pypde_plugin  # noqa: F401  (registers "hip")
    from pde.tools.misc import module_available

    if not module_available("numba"):
        # the reference's default backend is numba; where a generic test computes its yardstick without naming a
        # backend, the reference's scipy operators take its place in this container
        pde.config["default_backend"] = "scipy"
        # ... and where it names numba explicitly (`pde.PDE` on the numpy backend takes its operators from there,
        # pde/pdes/pde.py:349-358), the name resolves to the reference's scipy operators as well
        from pde.backends import backend_registry, get_backend

        backend_registry._backends["numba"] = get_backend("scipy")
        # ... and `numba.typed.Dict`, the container of the operators' `bc_args` on that path (pde/pdes/pde.py:469-482), is a dict
        import importlib.machinery
        import types

        nb, typed = types.ModuleType("numba"), types.ModuleType("numba.typed")
        nb.__spec__ = importlib.machinery.ModuleSpec("numba", None)
        typed.__spec__ = importlib.machinery.ModuleSpec("numba.typed", None)
        typed.Dict = dict
        nb.typed = typed
        sys.modules["numba"], sys.modules["numba.typed"] = nb, typed


def pytest_sessionfinish(session, exitstatus):
    import refpath

    if refpath.REAL:
        refpath.log_loaded_libraries("reference-suite child")


def pytest_unconfigure(config):
    global _shim_ctx
    if _shim_ctx is not None:
        _shim_ctx.__exit__(None, None, None)
        _shim_ctx = None


@pytest.hookimpl(hookwrapper=True)
def pytest_make_collect_report(collector):
    """`tests/pdes/test_pde_class.py` imports numba at module level for ONE numba-only test; the hip-parametrised tests of
    that file do not touch it.  While such a module is being imported an empty stand-in lets the import succeed when numba is
    not installed (removed again right away: py-pde must keep seeing numba as absent)."""
    import importlib.util
    import types

    stub = False
    if isinstance(collector, pytest.Module) and "numba" not in sys.modules and importlib.util.find_spec("numba") is None:
        sys.modules["numba"] = types.ModuleType("numba")
        stub = True
    try:
        yield
    finally:
        if stub:
            sys.modules.pop("numba", None)


@pytest.hookimpl(tryfirst=True)
def pytest_generate_tests(metafunc):
    """Runs before pytest resolves the ``parametrize`` marks: the marks hold the module's list OBJECTS, so editing
    them in place re-parametrises every generic test with the hip backend."""
    mod = metafunc.module
    # tests the reference parametrises over an explicit ["numpy", "numba"] list: the hip backend takes their place too
    for mark in metafunc.definition.iter_markers("parametrize"):
        if len(mark.args) >= 2 and mark.args[0] == "backend" and isinstance(mark.args[1], list):
            values = mark.args[1]
            if values and all(isinstance(v, str) for v in values) and "numpy" in values:
                values[:] = ["hip"]
    for name in BACKEND_LIST_NAMES:
        lst = getattr(mod, name, None)
        if isinstance(lst, list) and lst != ["hip"]:
            lst[:] = ["hip"]
    unsupported = getattr(mod, "NOT_SUPPORTED", None)
    if isinstance(unsupported, dict) and "hip" not in unsupported:
        # explicit steppers and, through make_pde_rhs, ScipySolver; the others must raise NotImplementedError
        from pde import solvers as S

        unsupported["hip"] = {getattr(S, n) for n in ("CrankNicolsonSolver", "ImplicitSolver") if hasattr(S, n)}


@pytest.fixture
def backend(request):
    from pde.backends import get_backend

    if not str(request.param).startswith("hip"):
        pytest.skip("only the hip backend is exercised by this run")
    return get_backend(request.param)


@pytest.fixture(name="rng")
def init_random_number_generators():
    return np.random.default_rng(0)


@pytest.fixture(autouse=True)
def _setup_and_teardown():
    old = np.seterr(all="raise", under="ignore")
    yield
    np.seterr(**old)
