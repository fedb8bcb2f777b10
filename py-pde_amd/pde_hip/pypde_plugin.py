"""Registration of the hip backend with a real py-pde installation.

Usage (py-pde installed, libpdehip built)::

    import pde
    import pde_hip.pypde_plugin          # registers backend "hip" with pde.backends.backend_registry
    field.laplace(bc, backend="hip")
    eq.solve(state, t_range=10, dt=0.1, backend="hip", solver="euler")
    eq.solve(..., backend="hip:2")        # device 2

Mechanism (``pde/backends/registry.py:57-84``, ``:101-230``; pattern of
``pde/backends/torch/__init__.py:11-20``): ``backend_registry.register_package("hip",
"pde_hip.pypde_plugin", config=[...])`` declares the package lazily; importing this module defines
``HipBackend(HipBackendMixin, pde.backends.base.BackendBase)``, registers the Cartesian operators on
it and calls ``backend_registry.register_class("hip", HipBackend)``.  The mixin duck-types py-pde's
grids, ``BoundariesList`` and PDE / solver objects, so no py-pde class is re-implemented here.

Importing this module without py-pde raises ``ImportError``.  Constructing the backend never touches
the HIP runtime (``grid.operators`` instantiates every registered backend and only tolerates
``ImportError``, ``pde/backends/registry.py:241-245``): the first compute call selects the device and
raises ``RuntimeError`` when the library or a GPU is missing — there is no CPU fallback.
"""

from __future__ import annotations

try:
    import pde  # noqa: F401
    from pde.backends import backend_registry
    from pde.backends.base import BackendBase
    from pde.grids.cartesian import CartesianGrid
    from pde.tools.config import Parameter
except ImportError as err:  # pragma: no cover - exercised only without py-pde
    msg = "pde_hip.pypde_plugin needs py-pde (`import pde` failed)"
    raise ImportError(msg) from err

import numpy as np

from . import operators as _operators
from .backend import HipBackendMixin
from .device import DeviceArray

DEFAULT_CONFIG = {
    "device": Parameter(value=0, cls=int, description="Index of the HIP device (MI355X) the backend runs on; `hip:<n>` overrides it."),
    "resident_state": Parameter(
        value=True,
        cls=bool,
        description="Keep the state on the device between the calls of one stepper: tracker interrupts only download it, "
        "and it is uploaded again only when the host copy was modified in between.",
    ),
}


class HipBackend(HipBackendMixin, BackendBase):
    """py-pde backend running the finite-difference / explicit-stepper hot path on MI355X."""

    def __init__(self, config=None, *, name: str = "hip", device: int | None = None):
        BackendBase.__init__(self, config, name=name)
        if device is None and "device" in self.config:
            device = int(self.config["device"])
        self._hip_init(device)

    @classmethod
    def from_args(cls, config, args: str = "", *, name: str | None = None):
        """``get_backend("hip:2")`` selects device 2 (pde/backends/registry.py:164-186)."""
        try:
            device = int(args) if args else None
        except ValueError:
            msg = f"hip backend: device index expected after the colon, got `{args}`"
            raise ValueError(msg) from None
        return cls(config, name=name or (f"hip:{args}" if args else "hip"), device=device)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(name={self.name!r}, device={self.device})"

    # py-pde hands native <-> numpy conversion only the array; the geometry is taken from the
    # operator/stepper closures, so plain arrays are accepted where the grid is implied
    def native_to_numpy(self, value):
        if isinstance(value, DeviceArray):
            return value.get_valid(stream=self.stream)
        return value

    def _apply_operator(self, func, *values: np.ndarray, out: np.ndarray, **kwargs) -> None:
        HipBackendMixin._apply_operator(self, func, *values, out=out, grid=getattr(func, "grid", None), **kwargs)


_operators.register_all(HipBackend, CartesianGrid)


def register() -> None:
    """(Re-)register the backend class with py-pde's registry; idempotent."""
    if "hip" not in backend_registry._packages:
        backend_registry.register_package("hip", __name__, config=DEFAULT_CONFIG)
    backend_registry.register_class("hip", HipBackend)


register()
