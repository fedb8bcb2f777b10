"""State residency between the calls of a stepper: :class:`ResidentState` and the field class that pulls the device copy on host access.  Split from
``backend.py`` in round 6 (no behaviour change).
"""

from __future__ import annotations

import ctypes as C
import inspect
import logging
import os
from collections import defaultdict
from typing import Any, Callable, NamedTuple

import numpy as np

from . import _abi
from ._lib import require_device
from .device import DeviceArray, DeviceBuffer, DeviceScalar, GridInfo, ptr_array

_logger = logging.getLogger("pde_hip.backend")


def _config_get(config, key: str, default):
    try:
        if config is not None and key in config:
            return config[key]
    except TypeError:
        pass
    return default


_DATA_ATTRIBUTES = frozenset({"data", "_data_valid", "_data_full", "_FieldBase__data_full"})
_SYNCED_CLASSES: dict[type, type] = {}


class ResidentState:
    """Link between a host field object and its device-resident copy (no full-field PCIe traffic per
    tracker interrupt; reference behaviour being replaced: ``pde/backends/torch/backend.py:654-662``).

    The field object handed to the stepper keeps its identity (the controller returns it, trackers receive it), but its
    class is swapped for a dynamic subclass whose data attributes (``data``, ``_data_full`` ...) first bring the host
    arrays up to date — one pinned-speed download — and then count as a possible modification, so the next stepper call
    uploads again.  Nothing else about the field changes; copies of it are ordinary fields.

    Limitation: synchronisation happens on ATTRIBUTE ACCESS.  A numpy view obtained earlier (``arr = state.data``
    kept by a tracker or by the caller) is not refreshed behind the holder's back while the run goes on — it shows the state of
    its last ``state.data`` access — and writes made through such a held view after that access are not seen.  Code that wants
    the reference's behaviour (the stepper updates the host array in place at every call) sets ``resident_state=False`` in the
    backend's configuration, which restores the upload / download per stepper call of ``pde/backends/torch/backend.py:654-662``.
    """

    def __init__(self, field, dev_state: DeviceArray, backend):
        self.dev_state, self.backend = dev_state, backend
        self.host_stale = False          # device is ahead of the host arrays
        self.host_touched = True         # host arrays may differ from the device copy (initially: never uploaded)
        self.downloads = self.uploads = 0

    @staticmethod
    def attach(field, dev_state: DeviceArray, backend) -> "ResidentState":
        link = field.__dict__.get("_hip_link")
        cls = type(field)
        base = getattr(cls, "_hip_base_class", cls)
        if base not in _SYNCED_CLASSES:
            _SYNCED_CLASSES[base] = _make_synced_class(base)
        if link is not None and link.dev_state is dev_state:
            if cls is base:              # a host access since the last call put the plain class back (before_host_access)
                field.__class__ = _SYNCED_CLASSES[base]
            return link
        if link is not None:             # a stepper of an earlier run: settle it first
            link.pull(field)
        link = ResidentState(field, dev_state, backend)
        field.__dict__["_hip_link"] = link
        link._field_ref = field
        if cls is base:
            field.__class__ = _SYNCED_CLASSES[base]
        return link

    def __reduce__(self):
        # a field that was read after the run is a plain py-pde object again but still carries this link in its `__dict__` (the next
        # stepper call picks it up): copies and pickles of the field get `None` in its place
        return (type(None), ())

    def __deepcopy__(self, memo):
        return None

    def _host_valid(self):
        field = self._field_ref
        base = getattr(type(field), "_hip_base_class", type(field))
        return base.data.fget(field) if isinstance(getattr(base, "data", None), property) else object.__getattribute__(field, "data")

    def push(self) -> None:
        if self.host_touched:
            field = self._field_ref
            field.__dict__["_hip_link"] = None            # plain access while we read the host arrays
            try:
                self.dev_state.set_valid(field.data, self.backend.stream)
            finally:
                field.__dict__["_hip_link"] = self
            self.host_touched, self.host_stale = False, False
            self.uploads += 1

    def device_advanced(self) -> None:
        self.host_stale = True

    def pull(self, field=None) -> None:
        """Bring the host arrays up to date (called on the first data access after the device advanced)."""
        field = self._field_ref if field is None else field
        if self.host_stale:
            self.host_stale = False
            field.__dict__["_hip_link"] = None
            try:
                self.dev_state.get_valid(out=field.data, stream=self.backend.stream)
            finally:
                field.__dict__["_hip_link"] = self
            self.downloads += 1

    def before_host_access(self) -> None:
        """First access to the data after a stepper call: the host arrays are current from here on and may be written, so nothing
        needs intercepting until the next stepper call - the field gets its own class back (``type(result) is pde.ScalarField`` once
        the result has been looked at; `attach` swaps the intercepting subclass in again)."""
        self.pull()
        self.host_touched = True
        field = self._field_ref
        base = getattr(type(field), "_hip_base_class", None)
        if base is not None:
            object.__dict__["__class__"].__set__(field, base)


def _make_synced_class(base: type) -> type:
    def __getattribute__(self, name):
        if name in _DATA_ATTRIBUTES:
            link = object.__getattribute__(self, "__dict__").get("_hip_link")
            if link is not None:
                link.before_host_access()
        return base.__getattribute__(self, name)

    def __reduce_ex__(self, protocol):
        # pickling / deepcopy: settle the data and present the plain class
        link = self.__dict__.pop("_hip_link", None)
        if link is not None:
            self.__dict__["_hip_link"] = None
            link.pull(self)
            del self.__dict__["_hip_link"]
        self.__class__ = base
        return base.__reduce_ex__(self, protocol)

    # `field.__class__` keeps answering with the field's own class: py-pde compares classes by identity before any binary
    # operation (`assert_field_compatible`, pde/fields/base.py:385-390) and builds copies from `self.__class__`; only
    # `type(field)` shows the intercepting subclass
    real_class = object.__dict__["__class__"]

    def _get_class(self):
        return base

    def _set_class(self, value):
        real_class.__set__(self, value)

    # py-pde registers every field subclass by NAME (pde/fields/base.py:77-88) to rebuild fields from stored attributes: the
    # registry must keep pointing at the real class
    import warnings

    registry = getattr(base, "_subclasses", None)
    previous = registry.get(base.__name__) if isinstance(registry, dict) else None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        synced = type(base.__name__, (base,), {"__getattribute__": __getattribute__, "__reduce_ex__": __reduce_ex__, "_hip_base_class": base,
                                               "__class__": property(_get_class, _set_class),
                                               "__module__": base.__module__, "__qualname__": base.__qualname__, "__doc__": base.__doc__})
    if previous is not None:
        registry[base.__name__] = previous
    return synced
