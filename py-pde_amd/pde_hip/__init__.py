"""pde_hip — MI355X (gfx950) backend for py-pde's finite-difference + explicit-stepper hot path.

Layers (bottom up):
  * ``lib/libpdehip.so``  hand-written HIP kernels behind the C ABI of ``include/pdehip.h``
  * ``_abi`` / ``_lib``    ctypes binding (loud failure when the library or the GPU is missing)
  * ``device``            device-resident ghost-padded arrays (the backend's native array type)
  * ``backend``           ``HipBackend`` — py-pde's ``BackendBase`` plugin surface
  * ``grids`` / ``boundaries`` / ``fields`` / ``pdes`` / ``solvers``  stand-alone mirror of the
    reference classes that call into the backend (same names, argument meaning, errors), used when
    py-pde itself is not installed; ``pypde_plugin`` registers the backend with a real py-pde.
  * ``mesh`` / ``distributed``  slab decomposition + halo exchange over RCCL (one process per GPU)

Importing this package never touches the GPU; the first compute call does and raises if the
library or a device is missing (there is no CPU fallback).
"""

from . import operators as _operators
from .backend import HipBackend, get_backend
from .boundaries import BoundariesList
from .fields import FieldCollection, ScalarField, Tensor2Field, VectorField
from .grids import CartesianGrid, UnitGrid
from .pdes import (PDE, AllenCahnPDE, CahnHilliardPDE, DiffusionPDE, KleinGordonPDE, KPZInterfacePDE, KuramotoSivashinskyPDE,
                   SwiftHohenbergPDE, WavePDE)
from .solvers import Controller, EulerSolver, ExplicitSolver, RungeKuttaSolver

_operators.register_all(HipBackend, CartesianGrid)

__all__ = [
    "PDE",
    "AllenCahnPDE",
    "KPZInterfacePDE",
    "KleinGordonPDE",
    "KuramotoSivashinskyPDE",
    "SwiftHohenbergPDE",
    "WavePDE",
    "BoundariesList",
    "CahnHilliardPDE",
    "CartesianGrid",
    "Controller",
    "DiffusionPDE",
    "EulerSolver",
    "FieldCollection",
    "ExplicitSolver",
    "HipBackend",
    "RungeKuttaSolver",
    "ScalarField",
    "Tensor2Field",
    "UnitGrid",
    "VectorField",
    "get_backend",
]
__version__ = "0.1.0"
