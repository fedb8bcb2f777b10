"""Slab-parallel explicit steppers: one process per GPU, halo exchange over RCCL (xGMI).

Replaces the reference's MPI path — ``ExplicitMPISolver`` (``pde/solvers/explicit_mpi.py:133-226``),
``_MPIBC`` ghost exchange (``pde/grids/boundaries/local.py:561-662``,
``pde/backends/numba_mpi/backend.py:30-194``) and the MAX all-reduce of the adaptive error
(``pde/backends/base.py:678-712``).  The reference exchanges faces with *blocking* sends inside every
right-hand side; here every time loop is ONE C call (``csrc/pdehip_slab_loops.h`` through
``pdehip_slab_*``) that enqueues kernels, ``ncclSend``/``ncclRecv`` groups and events without host work
per step or stage:

    comp stream : interior sweep (own layers only) ────────────────────────────┐ next step / pair …
    halo stream : boundary sweeps → send/recv of the new boundary layers (overlaps the interior sweep)

Axis-0 slabs keep every face contiguous (no pack kernel).  Data plane: libpdehip only (device memory,
streams, RCCL resolved by ``dlsym``).  Control plane: a few small host-side collectives at start-up and at
tracker interrupts (the RCCL unique id, agreeing on code paths, gathering results) through whatever
``torch.distributed`` process group the launcher initialised — ``gloo`` is enough, no tensor of the
simulation ever passes through torch.  There is no CPU fallback: the library needs a HIP device.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Any

import numpy as np

from . import _abi
from .device import DeviceBuffer
from .mesh import SlabMesh

FUSED_CH, FUSED_STAGE = 1, 2   # PDEHIP_SLAB_* of include/pdehip.h


# ---------------------------------------------------------------------------------------------
# control plane
# ---------------------------------------------------------------------------------------------
class TorchControl:
    """Host-side collectives of the slab stepper on a ``torch.distributed`` process group (objects only)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.active = dist.is_available() and dist.is_initialized()
        self.size = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0

    def broadcast(self, obj, src: int = 0):
        if self.size == 1:
            return obj
        box = [obj]
        self.dist.broadcast_object_list(box, src=src, group=self.group)
        return box[0]

    def allgather(self, obj) -> list:
        if self.size == 1:
            return [obj]
        out: list[Any] = [None] * self.size
        self.dist.all_gather_object(out, obj, group=self.group)
        return out

    def all_and(self, bits: int) -> int:
        """Bitwise AND over all ranks: code paths every rank can take (decided collectively)."""
        res = -1
        for b in self.allgather(int(bits)):
            res &= b
        return res

    def barrier(self) -> None:
        if self.size > 1:
            self.dist.barrier(group=self.group)

    # raw arrays (results of a run: no pickles, straight out of / into numpy memory); `bytes_sent` counts what this rank put on the wire
    bytes_sent = 0

    def _tensor(self, arr: np.ndarray):
        import torch

        return torch.from_numpy(arr.reshape(-1).view(np.uint8))

    def send_array(self, arr: np.ndarray, dst: int) -> None:
        arr = np.ascontiguousarray(arr)
        self.dist.send(self._tensor(arr), dst=dst, group=self.group)
        self.bytes_sent += arr.nbytes

    def recv_array(self, buf: np.ndarray, src: int) -> None:
        """Into ``buf`` (C-contiguous, writable)."""
        self.dist.recv(self._tensor(buf), src=src, group=self.group)

    def bcast_array(self, buf: np.ndarray, src: int) -> None:
        """``buf`` (C-contiguous) of rank ``src`` to every rank."""
        self.dist.broadcast(self._tensor(buf), src=src, group=self.group)
        if self.rank == src:
            self.bytes_sent += buf.nbytes * (self.size - 1)


class SerialControl:
    """World size 1 without torch."""

    size, rank = 1, 0

    def broadcast(self, obj, src: int = 0):
        return obj

    def allgather(self, obj) -> list:
        return [obj]

    def all_and(self, bits: int) -> int:
        return int(bits)

    def barrier(self) -> None:
        pass

    bytes_sent = 0


def default_control():
    try:
        import torch.distributed as dist
    except ImportError:   # pragma: no cover
        return SerialControl()
    if dist.is_available() and dist.is_initialized():
        return TorchControl()
    return SerialControl()


def agree_on_environment(control, names) -> None:
    """The schedule switches the C loops read from the environment (``PDEHIP_SLAB_*``, ``PDEHIP_BLOCK2_*``) change WHICH messages a rank
    sends and in WHICH order; RCCL pairs sends and receives by order only, so ranks with different settings would exchange the wrong
    layers without any error.  Every rank publishes its settings; any difference is an error on every rank."""
    mine = tuple(os.environ.get(n, "") for n in names)
    everyone = control.allgather(mine)
    if any(e != everyone[0] for e in everyone):
        msg = "the ranks disagree on " + ", ".join(names) + ": " + "; ".join(f"rank {r}: {e}" for r, e in enumerate(everyone))
        raise RuntimeError(msg)


def gather_parts(control, local: np.ndarray, boxes, out_shape, root: int | None = None, out: np.ndarray | None = None):
    """Assemble the parts of all ranks into the global array WITHOUT pickles and without an N-fold funnel (VERDICT r5 "next" #6; the thing to
    beat: ``GridMesh.combine_field_data_mpi``, pde/grids/_mesh.py:593-615, which gathers pickled sub-arrays on the main node).

    ``boxes[r]``: index tuple of rank r's part in the global array (leading component axes whole).  ``root`` = a rank: only that rank receives
    (point-to-point, each part crosses the control plane ONCE: total traffic = the field minus the root's own part); the others get ``None``
    back (and, given ``out``, their own part written into it).  ``root`` = None: every rank receives everything (one raw broadcast per part - what trackers on every rank need).  Parts land directly
    in the memory of the result where that is contiguous (axis-0 slabs of a scalar field), else through one temporary per part."""
    size, rank = control.size, control.rank
    local = np.ascontiguousarray(local)
    if size == 1:
        if out is None:
            return local.reshape(out_shape)
        out[...] = local.reshape(out_shape)
        return out
    if root is not None and rank != root:
        if out is not None:
            out[boxes[rank]] = local      # the caller's copy of the field keeps this rank's own part up to date
        control.send_array(local, root)
        return None
    if out is None:
        out = np.empty(out_shape, dtype=local.dtype)
    for r in range(size):
        view = out[boxes[r]]
        if r == rank:
            view[...] = local
            if root is None:
                control.bcast_array(local, r)
            continue
        direct = view.flags.c_contiguous and view.flags.writeable
        buf = view if direct else np.empty(view.shape, dtype=out.dtype)
        if root is None:
            control.bcast_array(buf, r)
        else:
            control.recv_array(buf, r)
        if not direct:
            view[...] = buf
    return out


def rccl_library_path() -> str:
    """The librccl.so libpdehip resolves with dlsym: the copy torch ships (one RCCL per process when torch's own
    RCCL backend is also in use), else the system one; ``PDEHIP_RCCL`` overrides."""
    env = os.environ.get("PDEHIP_RCCL")
    if env:
        return env
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
        if os.path.exists(cand):
            return cand
    return "librccl.so"


# ---------------------------------------------------------------------------------------------
# device arrays of one slab
# ---------------------------------------------------------------------------------------------
class SlabArray:
    """A slab array (ghost layer, n own layers, ghost layer) with ONE SPARE LAYER of memory before and after, so the same
    memory is also the array with two halo layers per side of the two-level sweeps (``include/pdehip.h``)."""

    def __init__(self, nelems: int, layer_pitch: int, itemsize: int):
        self.nbytes = (nelems + 2 * layer_pitch) * itemsize
        self._buffer = DeviceBuffer(self.nbytes)
        self.ptr = self._buffer.ptr + layer_pitch * itemsize

    @property
    def ext_ptr(self) -> int:
        return self._buffer.ptr


def device_bc_program(lib, kind, faces_c, faces_mu, grid_ref):
    """The device program (``pdehip_rhs_t::bc_program``) that rewrites the faces of a slab / block which depend on time or read
    the field before every right-hand side of the C loops, or None.  ``grid_ref``: the local ``pdehip_grid_t``."""
    from .bc_expr import program_for

    if kind == _abi.RHS_CAHN_HILLIARD and getattr(faces_mu, "reads_value", False):
        # the potential exists only between the two halves of the fused right-hand side (single GPU: expression form instead)
        msg = "decomposed stepping: conditions on the chemical potential that depend non-linearly on it are not supported"
        raise NotImplementedError(msg)
    return program_for(lib, [faces_c, faces_mu], grid_ref)


# ---------------------------------------------------------------------------------------------
# the slab stepper
# ---------------------------------------------------------------------------------------------
class SlabStepper:
    """Explicit Euler / RK4 / RKF45 for Diffusion and Cahn–Hilliard on an axis-0 slab decomposition."""

    def __init__(self, eq, grid, dtype=np.float64, *, control=None, device: int | None = None, force_exchange: bool = False):
        from ._lib import require_device

        self.control = control if control is not None else default_control()
        self.size, self.rank = self.control.size, self.control.rank
        self.lib = require_device(device)
        self.eq, self.grid, self.dtype = eq, grid, np.dtype(dtype)
        self.mesh = SlabMesh(grid, self.size, self.rank)
        self.n = self.mesh.n_local
        self.g = _abi.make_grid(self.mesh.local_shape, grid.discretization, self.dtype)
        lay = (C.c_int64 * 8)()
        self.lib.layout(C.byref(self.g), lay)
        self.comp_elems, self.layer_pitch = int(lay[2]), int(lay[7])
        self.nelems = self.comp_elems + int(lay[6])
        self.itemsize = self.dtype.itemsize
        self.stream = C.c_void_p()
        self.lib.stream_create(C.byref(self.stream))
        # neighbours; world size 1 + periodic axis 0 can be forced through the exchange path (exchange with itself)
        self.lower, self.upper = self.mesh.lower, self.mesh.upper
        if force_exchange and self.size == 1 and grid.periodic[0]:
            self.lower = self.upper = 0
        self.exchanging = self.lower is not None or self.upper is not None
        self._lo = -1 if self.lower is None else int(self.lower)
        self._up = -1 if self.upper is None else int(self.upper)
        # right-hand side description
        self.kind, self.param, bc_c, bc_mu = self._describe(eq, grid)
        self.faces_c = self.mesh.slab_faces(bc_c, force_exchange=force_exchange)
        self.faces_mu = self.faces_c if bc_mu is bc_c else self.mesh.slab_faces(bc_mu, force_exchange=force_exchange)
        self._bufs: dict[str, SlabArray] = {}
        self.rhs = _abi.RHS()
        self.rhs.kind, self.rhs.param = self.kind, self.param
        self.faces_c.copy_into(self.rhs.bc_c)
        self.faces_mu.copy_into(self.rhs.bc_mu)
        if self.kind == _abi.RHS_CAHN_HILLIARD:
            self.rhs.scratch_mu = self.buf("mu").ptr
        # faces that depend on time or read the field: rewritten on the device before every right-hand side of the C loops
        self.bc_program = device_bc_program(self.lib, self.kind, self.faces_c, self.faces_mu, C.byref(self.g))
        if self.bc_program is not None:
            self.rhs.bc_program = self.bc_program.ptr
        self.err = DeviceBuffer(8)
        # communicator: libpdehip's own RCCL communicator; the 128-byte id travels over the control plane
        self.comm = None
        if self.exchanging or self.size > 1:
            path = rccl_library_path().encode()
            uid = C.create_string_buffer(128)
            if self.rank == 0:
                self.lib.comm_unique_id(path, uid)
            uid = C.create_string_buffer(self.control.broadcast(bytes(uid.raw)), 128)
            self.comm = C.c_void_p()
            self.lib.comm_create(path, uid, self.rank, self.size, C.byref(self.comm))
        # code paths: every rank asks its library, the answers are ANDed over all ranks, so that nobody exchanges two
        # layers while its neighbour exchanges one
        agree_on_environment(self.control, ("PDEHIP_SLAB_EULER2", "PDEHIP_SLAB_EULER4", "PDEHIP_SLAB_DEEP_MODE", "PDEHIP_SLAB_THICK", "PDEHIP_SLAB_RIM"))
        local = C.c_int(0)
        self.lib.slab_flags_supported(C.byref(self.g), C.byref(self.rhs), self._lo, self._up, C.byref(local))
        e2 = C.c_int(0)
        if self.kind == _abi.RHS_DIFFUSION and self.exchanging and min(self.mesh.counts) >= 4:
            self.lib.slab_euler2_supported(C.byref(self.g), C.byref(self.rhs), C.byref(e2))
        # ... and four steps per exchange (four halo layers per side, csrc/pdehip_slab_loops.h: euler4_run) with >= 8 layers on every rank
        e4 = C.c_int(0)
        if e2.value and min(self.mesh.counts) >= 8 and os.environ.get("PDEHIP_SLAB_EULER4", "1") != "0":
            self.lib.slab_euler4_supported(C.byref(self.g), C.byref(self.rhs), C.byref(e4))
        if os.environ.get("PDEHIP_SLAB_EULER2", "1") == "0" or self.bc_program is not None:
            e2.value = e4.value = 0     # (faces that change from step to step: one step per sweep)
        if self.kind == _abi.RHS_CAHN_HILLIARD and min(self.mesh.counts) < 2:
            local.value &= ~FUSED_CH
        agreed = self.control.all_and(local.value | (4 if e2.value else 0) | (8 if e4.value else 0))
        self.flags = agreed & (FUSED_CH | FUSED_STAGE)
        if self.kind == _abi.RHS_CAHN_HILLIARD and not self.flags & FUSED_CH:
            self.flags = 0   # the stage epilogue of Cahn-Hilliard rides on the two-level sweep
        self._euler2 = bool(agreed & 4)
        self._euler4 = bool(agreed & 8) and self._euler2

    @staticmethod
    def _describe(eq, grid):
        from .backend import pde_kind

        name = pde_kind(eq)
        if name == "DiffusionPDE":
            bc = grid.get_boundary_conditions(eq.bc, rank=0)
            return _abi.RHS_DIFFUSION, float(eq.diffusivity), bc, bc
        if name == "CahnHilliardPDE":
            return (_abi.RHS_CAHN_HILLIARD, float(eq.interface_width), grid.get_boundary_conditions(eq.bc_c, rank=0),
                    grid.get_boundary_conditions(eq.bc_mu, rank=0))
        if name == "PDE":
            # expression PDEs that map onto the fused right-hand sides (BASELINE config 5)
            from .backend import _match_expression_rhs, pde_bc_for, pde_expression

            rhs = dict(eq.rhs)
            match = None
            if len(rhs) == 1:
                (var,) = rhs
                match = _match_expression_rhs(pde_expression(eq, var), var, dict(getattr(eq, "consts", {}) or {}))
            if match is None:
                msg = "slab stepper supports expression PDEs of the Diffusion / Cahn-Hilliard form"
                raise NotImplementedError(msg)
            # one condition per operator name, inner and outer laplace alike (pde/pdes/pde.py:329-343)
            bc = grid.get_boundary_conditions(pde_bc_for(eq, var, "laplace"), rank=0)
            return match[0], match[1], bc, bc
        msg = f"slab stepper has no fused right-hand side for {name}"
        raise NotImplementedError(msg)

    # --- buffers ---------------------------------------------------------------------------------
    def buf(self, name: str) -> SlabArray:
        if name not in self._bufs:
            self._bufs[name] = SlabArray(self.nelems, self.layer_pitch, self.itemsize)
        return self._bufs[name]

    def _work(self, names) -> C.Array:
        arr = (C.c_void_p * len(names))()
        for i, n in enumerate(names):
            arr[i] = self.buf(n).ptr
        return arr

    def synchronize(self) -> None:
        self.lib.stream_synchronize(self.stream)

    # --- halo exchange / reductions (building blocks, also used by tests) ---------------------------------------
    def exchange(self, buf: SlabArray) -> None:
        """Fill the ghost layers of ``buf`` from the neighbours (one layer per side)."""
        if self.exchanging:
            self.lib.halo_exchange(self.comm, C.byref(self.g), buf.ptr, self._lo, self._up, self.stream)

    def sync_max(self, value: float) -> float:
        """MAX over all ranks of one value, NaN wins (``make_mpi_synchronizer``, pde/backends/base.py:678-712)."""
        host = C.c_double(value)
        self.lib.memcpy_h2d(self.err.ptr, C.addressof(host), 8, self.stream)
        if self.comm is not None:
            self.lib.allreduce_max(self.comm, self.err.ptr, self.stream)
        self.lib.memcpy_d2h(C.addressof(host), self.err.ptr, 8, self.stream)
        self.synchronize()
        return host.value

    # --- time loops: one C call each ---------------------------------------------------------------------------
    def rhs_scaled(self, y: SlabArray, k_out: SlabArray, dt: float, t: float = 0.0) -> None:
        self.rhs.t = float(t)
        self.lib.slab_rhs_scaled(self.comm, C.byref(self.g), C.byref(self.rhs), self._lo, self._up, self.flags, y.ptr, k_out.ptr, dt, self.stream)

    def euler_steps(self, cur: SlabArray, nxt: SlabArray, dt: float, nsteps: int, t0: float = 0.0) -> SlabArray:
        """``nsteps`` Euler steps from time ``t0`` ping-ponging cur/nxt; returns the array holding the result."""
        res = C.c_void_p()
        self.rhs.t = float(t0)
        g, rhs = C.byref(self.g), C.byref(self.rhs)
        if not self.exchanging:
            self.lib.euler_run(g, rhs, cur.ptr, nxt.ptr, dt, nsteps, C.byref(res), self.stream)
        elif self.kind != _abi.RHS_DIFFUSION or self.bc_program is not None:
            self.lib.slab_euler_sweeps(self.comm, g, rhs, self._lo, self._up, self.flags, cur.ptr, nxt.ptr, dt, nsteps, C.byref(res), self.stream)
        else:
            run = self.lib.slab_euler_run
            if self._euler2 and nsteps >= 2:
                run = self.lib.slab_euler4_run if self._euler4 and nsteps >= 4 else self.lib.slab_euler2_run
            run(self.comm, g, rhs, self._lo, self._up, cur.ptr, nxt.ptr, dt, nsteps, C.byref(res), self.stream)
        return cur if res.value == cur.ptr else nxt

    def rk4_steps(self, y: SlabArray, dt: float, nsteps: int, t0: float = 0.0) -> None:
        work = self._work(["k1", "k2", "k3", "k4", "tmp"])
        self.rhs.t = float(t0)
        self.lib.slab_rk4_run(self.comm, C.byref(self.g), C.byref(self.rhs), self._lo, self._up, self.flags, y.ptr, work, dt, nsteps, self.stream)

    def rkf45_run(self, cur: SlabArray, nxt: SlabArray, ctl: _abi.Adaptive) -> SlabArray:
        """Adaptive RKF45 from ``ctl.t_start`` to ``ctl.t_end`` (accept/reject, controller, MAX all-reduce all in C)."""
        work = self._work(["k1", "k2", "k3", "k4", "k5", "k6", "tmp"])
        res = C.c_void_p()
        self.lib.slab_rkf45_run(self.comm, C.byref(self.g), C.byref(self.rhs), self._lo, self._up, self.flags, cur.ptr, nxt.ptr, work,
                                self.err.ptr, C.byref(ctl), C.byref(res), self.stream)
        return cur if res.value == cur.ptr else nxt

    def euler_adaptive_run(self, cur: SlabArray, nxt: SlabArray, ctl: _abi.Adaptive) -> SlabArray:
        """The reference's adaptive Euler loop (rate carried between attempts, ``pde/backends/numba/_solvers.py:322-466``) from
        ``ctl.t_start`` to ``ctl.t_end`` in one C call, incl. the halo exchanges and the MAX all-reduce of the error."""
        work = self._work(["k1", "tmp", "k2"])   # rate, half step (a sweep input), slope scratch
        res = C.c_void_p()
        self.lib.slab_euler_adaptive_run(self.comm, C.byref(self.g), C.byref(self.rhs), self._lo, self._up, self.flags, cur.ptr, nxt.ptr, work,
                                         self.err.ptr, C.byref(ctl), C.byref(res), self.stream)
        return cur if res.value == cur.ptr else nxt

    # --- user level --------------------------------------------------------------------------------------
    def set_local(self, buf: SlabArray, local_valid: np.ndarray) -> None:
        host = np.ascontiguousarray(local_valid, dtype=self.dtype)
        stage = DeviceBuffer(host.nbytes)
        self.lib.memcpy_h2d(stage.ptr, host.ctypes.data, host.nbytes, self.stream)
        self.lib.valid_to_full(C.byref(self.g), 1, stage.ptr, buf.ptr, self.stream)
        self.synchronize()

    def scatter(self, global_valid: np.ndarray) -> SlabArray:
        """Upload this rank's block of a (replicated) global initial state; returns the array."""
        buf = self.buf("state_a")
        self.set_local(buf, self.mesh.extract(global_valid))
        return buf

    def gather_local(self, buf: SlabArray) -> np.ndarray:
        shape = self.mesh.local_shape
        host = np.empty(shape, dtype=self.dtype)
        stage = DeviceBuffer(host.nbytes)
        self.lib.full_to_valid(C.byref(self.g), 1, buf.ptr, stage.ptr, self.stream)
        self.lib.memcpy_d2h(host.ctypes.data, stage.ptr, host.nbytes, self.stream)
        self.synchronize()
        return host

    def get_hostfull(self, buf: SlabArray) -> np.ndarray:
        """The slab incl. its ghost layers in the reference's compact full layout (tests)."""
        shape = tuple(s + 2 for s in self.mesh.local_shape)
        host = np.empty(shape, dtype=self.dtype)
        stage = DeviceBuffer(host.nbytes)
        self.lib.full_to_hostfull(C.byref(self.g), 1, buf.ptr, stage.ptr, self.stream)
        self.lib.memcpy_d2h(host.ctypes.data, stage.ptr, host.nbytes, self.stream)
        self.synchronize()
        return host

    def gather(self, buf: SlabArray, root: int | None = None, out: np.ndarray | None = None):
        """The global valid array from the slabs of all ranks: on every rank (``root`` None: tracker interrupts of runs whose trackers live
        on every rank) or on rank ``root`` only (the others return None) - raw buffers straight into the result, see :func:`gather_parts`."""
        offsets = np.concatenate([[0], np.cumsum(self.mesh.counts)])
        boxes = [(slice(int(offsets[r]), int(offsets[r + 1])),) for r in range(self.size)]
        return gather_parts(self.control, self.gather_local(buf), boxes, tuple(self.grid.shape), root, out)

    def solve(self, global_valid: np.ndarray, t_range: float, dt: float | None, solver: str = "euler", *, tolerance: float = 1e-4,
              dt_min: float = 1e-10, dt_max: float = 1e10, root: int | None = None) -> tuple[np.ndarray, dict[str, Any]]:
        """Slab-parallel twin of ``eq.solve(...)`` with ``tracker=None``; returns (global final state, info) - the state on every rank, or
        (``root`` = a rank) on that rank only, None on the others (each part then crosses the control plane once: :func:`gather_parts`)."""
        cur = self.scatter(global_valid)
        nxt = self.buf("state_b")
        info: dict[str, Any] = {"steps": 0, "world_size": self.size, "flags": self.flags, "two_steps_per_sweep": self._euler2,
                                "steps_per_exchange": 4 if self._euler4 else (2 if self._euler2 else 1)}
        if dt is not None:
            steps = max(1, round(t_range / dt))
            if solver == "euler":
                cur = self.euler_steps(cur, nxt, dt, steps)
            elif solver == "runge-kutta":
                self.rk4_steps(cur, dt, steps)
            else:
                msg = f"slab stepper does not support solver {solver}"
                raise NotImplementedError(msg)
            info.update(steps=steps, dt=dt, t_final=(steps - 1) * dt + dt)
        else:
            if solver not in ("runge-kutta", "euler"):
                msg = "adaptive slab stepping is implemented for runge-kutta (RKF45) and euler"
                raise NotImplementedError(msg)
            ctl = _abi.Adaptive()
            ctl.t_start, ctl.t_end, ctl.dt, ctl.tolerance, ctl.dt_min, ctl.dt_max = 0.0, float(t_range), 1e-3, tolerance, dt_min, dt_max
            cur = self.rkf45_run(cur, nxt, ctl) if solver == "runge-kutta" else self.euler_adaptive_run(cur, nxt, ctl)
            info.update(steps=int(ctl.steps), attempts=int(ctl.attempts), dt=ctl.dt, t_final=ctl.t_last, dt_statistics=_abi.adaptive_statistics(ctl))
        self.synchronize()
        return self.gather(cur, root=root), info

    def close(self) -> None:
        """Release the communicator (with its scratch arrays) and the stream (the stepper cannot be used afterwards; its arrays are freed
        with the object).  Without a communicator the loops used the process-wide scratch: handed back too (allocated again on demand)."""
        if self.comm is not None:
            self.synchronize()
            self.lib.comm_destroy(self.comm)
            self.comm = None
        elif getattr(self, "stream", None) is not None:
            try:
                self.lib.release_scratch()
            except Exception:  # noqa: BLE001 - see below
                pass
        if getattr(self, "stream", None) is not None:
            stream, self.stream = self.stream, None
            try:
                self.lib.stream_synchronize(stream)
                self.lib.stream_destroy(stream)
            except Exception:  # noqa: BLE001 - releasing a resource at the end of a run must not turn a finished run into a failure
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


# ---------------------------------------------------------------------------------------------
# block decomposition: one box of the grid per process (csrc/pdehip_block_loops.h)
# ---------------------------------------------------------------------------------------------
class BlockStepper:
    """Explicit Euler / RK4 / adaptive RKF45 for Diffusion and Cahn-Hilliard on a BLOCK decomposition of the grid, e.g.
    2 x 2 x 2 for 512^3 on the 8 GPUs of a node (the reference's ``GridMesh`` with its "auto" rule, pde/grids/_mesh.py:59-93, under
    ``ExplicitMPISolver``, pde/solvers/explicit_mpi.py:133-226).  Every run is ONE C call (``pdehip_block_run``): before each
    right-hand side the six faces of its input travel to the face neighbours in one RCCL group (faces normal to the fast axes
    through packed staging buffers); the adaptive loop's error is MAX-reduced over all ranks.

    Slabs (:class:`SlabStepper`) keep every face contiguous and run the two-steps-per-sweep kernels; blocks cut the bytes per
    xGMI link by four at 8 ranks (0.5 MiB per face instead of 2 MiB for 512^3) and take one right-hand side per sweep.
    """

    def __init__(self, eq, grid, dtype=np.float64, *, dims=None, control=None, device: int | None = None, force_exchange: bool = False):
        from ._lib import require_device
        from .device import DeviceArray, GridInfo
        from .mesh import BlockMesh, block_decomposition

        self.control = control if control is not None else default_control()
        self.size, self.rank = self.control.size, self.control.rank
        self.lib = require_device(device)
        self.eq, self.grid, self.dtype = eq, grid, np.dtype(dtype)
        self.dims = [int(d) for d in (dims if dims is not None else block_decomposition(grid.shape, self.size))]
        if int(np.prod(self.dims)) != self.size:
            msg = f"decomposition {self.dims} needs {int(np.prod(self.dims))} ranks, the job has {self.size}"
            raise ValueError(msg)
        self.mesh = BlockMesh(grid, self.dims, self.rank, force_exchange=force_exchange)
        self.info = GridInfo(self.mesh.local_shape, grid.discretization, self.dtype)
        self._array = lambda: DeviceArray(self.info)
        self.stream = C.c_void_p()
        self.lib.stream_create(C.byref(self.stream))
        self.kind, self.param, bc_c, bc_mu = SlabStepper._describe(eq, grid)
        self.faces_c = self.mesh.block_faces(bc_c)
        self.faces_mu = self.faces_c if bc_mu is bc_c else self.mesh.block_faces(bc_mu)
        self.rhs = _abi.RHS()
        self.rhs.kind, self.rhs.param = self.kind, self.param
        self.faces_c.copy_into(self.rhs.bc_c)
        self.faces_mu.copy_into(self.rhs.bc_mu)
        self._bufs: dict[str, Any] = {}
        if self.kind == _abi.RHS_CAHN_HILLIARD:
            self.rhs.scratch_mu = self.buf("mu").ptr
        self.bc_program = device_bc_program(self.lib, self.kind, self.faces_c, self.faces_mu, self.info.ref)
        if self.bc_program is not None:
            self.rhs.bc_program = self.bc_program.ptr
        self.err = DeviceBuffer(8)
        self.nb6 = self.mesh.nb6
        self.exchanging = any(v >= 0 for v in self.nb6)
        self.comm = None
        if self.exchanging or self.size > 1:
            path = rccl_library_path().encode()
            uid = C.create_string_buffer(128)
            if self.rank == 0:
                self.lib.comm_unique_id(path, uid)
            uid = C.create_string_buffer(self.control.broadcast(bytes(uid.raw)), 128)
            self.comm = C.c_void_p()
            self.lib.comm_create(path, uid, self.rank, self.size, C.byref(self.comm))
        # the stage epilogue of the diffusion sweeps: every rank asks its library for its block, the answers are ANDed
        local = C.c_int(0)
        self.lib.slab_flags_supported(self.info.ref, C.byref(self.rhs), -1, -1, C.byref(local))
        fuse = bool(local.value & FUSED_STAGE) and self.kind == _abi.RHS_DIFFUSION
        self.fuse_stage = bool(self.control.all_and(1 if fuse else 0))
        # The FAST loop (csrc/pdehip_block2_loops.h: two steps per sweep on the box, halos two layers deep incl. the edges in one message
        # per neighbouring rank, the exchange hidden behind the next sweep): diffusion with fixed Euler steps on a 3-D grid that is
        # periodic on every axis and not cut along the fastest one.  Decided for ALL ranks alike (PDEHIP_BLOCK2=0: the exact one-step loop).
        nd = len(grid.shape)
        self.cut = [int(self.dims[a] > 1 or (force_exchange and bool(grid.periodic[a]))) for a in range(nd)]
        agree_on_environment(self.control, ("PDEHIP_BLOCK2", "PDEHIP_BLOCK2_MODE", "PDEHIP_BLOCK2_DIRECT", "PDEHIP_BLOCK2_CUS"))
        fast = 0
        if (nd == 3 and all(grid.periodic) and self.kind == _abi.RHS_DIFFUSION and self.bc_program is None
                and os.environ.get("PDEHIP_BLOCK2", "1") != "0"):
            if self.cut[2] and force_exchange and self.dims[2] == 1 and os.environ.get("PDEHIP_PROBE_CUT_FASTEST", "0") != "1":
                self.cut[2] = 0          # (the probe on one device: the fastest axis keeps its periodic wrap unless asked otherwise)
            # (the faces of the uncut axes as the kernels see them: the periodic conditions of the whole grid)
            whole = _abi.RHS()
            whole.kind, whole.param = self.kind, self.param
            BlockMesh(grid, [1] * nd, 0).block_faces(bc_c).copy_into(whole.bc_c)
            for a in range(nd):
                for side in range(2):
                    if not self.cut[a]:
                        # index of the wrap inside the BOX: its last / first cell
                        whole.bc_c[2 * a + side].index1 = 0 if side else self.mesh.local_shape[a] - 1
            self._rhs2 = whole
            self._cut3 = (C.c_int * 3)(*self.cut)
            ok = C.c_int(0)
            self.lib.block2_supported(self.info.ref, C.byref(whole), self._cut3, C.byref(ok))
            fast = int(ok.value)
        self.block2 = bool(self.control.all_and(fast))
        if self.block2 and self.comm is None and any(self.cut):
            self.block2 = False

    def buf(self, name: str):
        if name not in self._bufs:
            self._bufs[name] = self._array()
        return self._bufs[name]

    def synchronize(self) -> None:
        self.lib.stream_synchronize(self.stream)

    def exchange(self, arr) -> None:
        """Fill the ghost layers of ``arr`` on every face that has a neighbour."""
        self.lib.block_exchange(self.comm, self.info.ref, self.nb6, arr.ptr, self.stream)

    def _run(self, scheme: int, y, ynew, work, dt: float, nsteps: int, ctl=None, t0: float = 0.0):
        from .device import ptr_array

        self.rhs.t = float(t0)
        res = C.c_void_p()
        self.lib.block_run(self.comm, self.info.ref, C.byref(self.rhs), self.nb6, int(self.fuse_stage), scheme, y.ptr,
                           None if ynew is None else ynew.ptr, None if not work else ptr_array(work), self.err.ptr, float(dt), int(nsteps),
                           None if ctl is None else C.byref(ctl), C.byref(res), self.stream)
        return y if res.value == y.ptr else ynew

    def euler_steps(self, cur, nxt, dt: float, nsteps: int, t0: float = 0.0):
        if self.block2 and nsteps >= 2:
            # pairs of steps through the fast loop (in place on `cur`), an odd last step through the one-step loop
            res = C.c_void_p()
            dims3, coords3 = (C.c_int * 3)(*self.dims), (C.c_int * 3)(*self.mesh.index)
            self.lib.block2_euler_run(self.comm, self.info.ref, C.byref(self._rhs2), dims3, coords3, self._cut3, cur.ptr, nxt.ptr, float(dt),
                                      int(nsteps - nsteps % 2), C.byref(res), self.stream)
            return self._run(0, cur, nxt, None, dt, 1, t0=t0 + (nsteps - 1) * dt) if nsteps % 2 else cur
        return self._run(0, cur, nxt, None, dt, nsteps, t0=t0)

    def rk4_steps(self, y, dt: float, nsteps: int, t0: float = 0.0) -> None:
        self._run(1, y, None, [self.buf(n) for n in ("k1", "k2", "k3", "k4", "tmp")], dt, nsteps, t0=t0)

    def rkf45_run(self, cur, nxt, ctl: _abi.Adaptive):
        return self._run(2, cur, nxt, [self.buf(n) for n in ("k1", "k2", "k3", "k4", "k5", "k6", "tmp")], 0.0, 0, ctl)

    def euler_adaptive_run(self, cur, nxt, ctl: _abi.Adaptive):
        """The reference's adaptive Euler loop (``pde/backends/numba/_solvers.py:322-466``) on the block decomposition."""
        return self._run(3, cur, nxt, [self.buf(n) for n in ("k1", "tmp", "k2")], 0.0, 0, ctl)

    # --- user level --------------------------------------------------------------------------------------
    def scatter(self, global_valid: np.ndarray):
        """Upload this rank's block of a (replicated) global initial state; returns the array."""
        return self.buf("state_a").set_valid(np.ascontiguousarray(self.mesh.extract(global_valid), dtype=self.dtype), self.stream)

    def gather_local(self, arr) -> np.ndarray:
        return arr.get_valid(stream=self.stream)

    def set_local(self, arr, local_valid: np.ndarray) -> None:
        """Upload this rank's block (the interface of :meth:`SlabStepper.set_local`)."""
        arr.set_valid(np.ascontiguousarray(local_valid, dtype=self.dtype), self.stream)

    def _boxes(self, lead: int = 0):
        from .mesh import BlockMesh

        boxes = []
        for r in range(self.size):
            m = BlockMesh(self.grid, self.dims, r)
            boxes.append((slice(None),) * lead + tuple(slice(lo, hi) for lo, hi in zip(m.lo, m.hi)))
        return boxes

    def gather(self, arr, root: int | None = None, out: np.ndarray | None = None):
        """The global valid array from the blocks of all ranks, on every rank or on rank ``root`` only (see :meth:`SlabStepper.gather`)."""
        return gather_parts(self.control, self.gather_local(arr), self._boxes(), tuple(self.grid.shape), root, out)

    def solve(self, global_valid: np.ndarray, t_range: float, dt: float | None, solver: str = "euler", *, tolerance: float = 1e-4,
              dt_min: float = 1e-10, dt_max: float = 1e10) -> tuple[np.ndarray, dict[str, Any]]:
        """Block-parallel twin of ``eq.solve(...)`` with ``tracker=None``; returns (global final state, info)."""
        cur, nxt = self.scatter(global_valid), self.buf("state_b")
        info: dict[str, Any] = {"steps": 0, "world_size": self.size, "decomposition": list(self.dims), "fused_stages": self.fuse_stage}
        if dt is not None:
            steps = max(1, round(t_range / dt))
            if solver == "euler":
                cur = self.euler_steps(cur, nxt, dt, steps)
            elif solver == "runge-kutta":
                self.rk4_steps(cur, dt, steps)
            else:
                msg = f"block stepper does not support solver {solver}"
                raise NotImplementedError(msg)
            info.update(steps=steps, dt=dt, t_final=(steps - 1) * dt + dt)
        else:
            if solver not in ("runge-kutta", "euler"):
                msg = "adaptive block stepping is implemented for runge-kutta (RKF45) and euler"
                raise NotImplementedError(msg)
            ctl = _abi.Adaptive()
            ctl.t_start, ctl.t_end, ctl.dt, ctl.tolerance, ctl.dt_min, ctl.dt_max = 0.0, float(t_range), 1e-3, tolerance, dt_min, dt_max
            cur = self.rkf45_run(cur, nxt, ctl) if solver == "runge-kutta" else self.euler_adaptive_run(cur, nxt, ctl)
            info.update(steps=int(ctl.steps), attempts=int(ctl.attempts), dt=ctl.dt, t_final=ctl.t_last, dt_statistics=_abi.adaptive_statistics(ctl))
        self.synchronize()
        return self.gather(cur), info

    def close(self) -> None:
        """Release the communicator (with its scratch arrays) and the stream (the stepper cannot be used afterwards; its arrays are freed
        with the object).  Without a communicator the loops used the process-wide scratch: handed back too (allocated again on demand)."""
        if self.comm is not None:
            self.synchronize()
            self.lib.comm_destroy(self.comm)
            self.comm = None
        elif getattr(self, "stream", None) is not None:
            try:
                self.lib.release_scratch()
            except Exception:  # noqa: BLE001 - see below
                pass
        if getattr(self, "stream", None) is not None:
            stream, self.stream = self.stream, None
            try:
                self.lib.stream_synchronize(stream)
                self.lib.stream_destroy(stream)
            except Exception:  # noqa: BLE001 - releasing a resource at the end of a run must not turn a finished run into a failure
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


# ---------------------------------------------------------------------------------------------
# any expression PDE on slabs / blocks
# ---------------------------------------------------------------------------------------------
def resolve_decomposition(dims, size: int) -> list[int]:
    """Blocks per axis with ``-1`` entries replaced by the ranks that are left (``GridMesh.from_grid``, pde/grids/_mesh.py:230-256: one
    axis may be given as -1)."""
    dims = [int(d) for d in dims]
    free = [a for a, d in enumerate(dims) if d == -1]
    if len(free) > 1:
        msg = "only one axis of the decomposition can be -1"
        raise ValueError(msg)
    if free:
        fixed = int(np.prod([d for d in dims if d != -1]))
        dims[free[0]] = max(1, size // max(1, fixed))
    return dims


def create_communicator(lib, control):
    """libpdehip's RCCL communicator over all ranks of ``control`` (the 128-byte id travels over the control plane)."""
    path = rccl_library_path().encode()
    uid = C.create_string_buffer(128)
    if control.rank == 0:
        lib.comm_unique_id(path, uid)
    uid = C.create_string_buffer(control.broadcast(bytes(uid.raw)), 128)
    comm = C.c_void_p()
    lib.comm_create(path, uid, control.rank, control.size, C.byref(comm))
    return comm


class DecomposedExpressionStepper:
    """Explicit Euler / RK4 / adaptive steppers for ANY expression PDE the hip backend evaluates with its run-time compiled passes
    (``pde.PDE`` with scalar fields, systems of them, the built-in classes in their expression form) on a slab or block
    decomposition - what ``ExplicitMPISolver`` does for every PDE in the reference (``pde/solvers/explicit_mpi.py:133-226``: each
    node evaluates the right-hand side on its sub-grid, ``_MPIBC`` exchanges the ghost cells of every operator's operand,
    ``pde/grids/boundaries/local.py:561-662``).

    One rank = one box.  The expression is planned once (``pde_hip/expr.py``); its passes run one by one on the box, and before
    a pass applies operators to an array, the ghost layers of THAT array travel (``pdehip_halo_exchange`` / ``pdehip_block_exchange``:
    the operand of a nested operator is an intermediate field, which is exchanged like the state).  Conditions - constants, arrays,
    expressions of time / position / the field - are lowered for the box with the coordinates of the whole grid
    (``SlabMesh.slab_faces`` / ``BlockMesh.block_faces``); coordinate arrays and array-valued constants are cut to the box.  The
    steppers are the Python-level twins of the C loops (``HipBackendMixin._make_expression_stepper``); the adaptive error is
    MAX-reduced over the ranks.  Everything runs on the null stream (conditions are refreshed there): in order, no overlap - the
    fast decomposed paths are the fused Diffusion / Cahn-Hilliard loops of :class:`SlabStepper`.  Integrals over the grid are summed over
    the ranks (equal to the serial value up to rounding)."""

    def __init__(self, eq, state, *, dims=None, control=None, device: int | None = None, force_exchange: bool = False):
        """``force_exchange`` (world size 1): periodic axes exchange with the box itself instead of keeping their periodic condition
        (the exchange path on one GPU: tests)."""
        from ._lib import require_device
        from .backend import HipBackendMixin
        from .device import GridInfo
        from .mesh import BlockMesh, block_decomposition

        self.control = control if control is not None else default_control()
        self.size, self.rank = self.control.size, self.control.rank
        self.lib = self._lib = require_device(device)
        # a stream of its own (round 4: the refresh of time-dependent conditions follows the consumers' stream; PDEHIP_DECOMP_NULL_STREAM=1:
        # the null stream of round 3)
        self.stream = None
        if os.environ.get("PDEHIP_DECOMP_NULL_STREAM", "0") != "1":
            self.stream = C.c_void_p()
            self.lib.stream_create(C.byref(self.stream))
        self.eq, self.grid = eq, state.grid
        grid = state.grid
        self.dtype = np.dtype(state.dtype)
        nd = len(grid.shape)
        requested = dims
        if isinstance(dims, str):      # "auto": the reference's rule (pde/grids/_mesh.py:59-93); "slab" / None: axis 0 only
            dims = block_decomposition(grid.shape, self.size) if dims == "auto" else None
        dims = resolve_decomposition(dims if dims is not None else [self.size] + [1] * (nd - 1), self.size)
        if int(np.prod(dims)) != self.size:
            msg = f"decomposition {dims} needs {int(np.prod(dims))} ranks, the job has {self.size}"
            raise ValueError(msg)
        self.dims = dims
        self.blocks = any(d > 1 for d in dims[1:]) or (force_exchange and self.size == 1 and nd > 1 and requested == "auto")
        self._force = bool(force_exchange and self.size == 1 and any(grid.periodic[: None if self.blocks else 1]))
        self.mesh = BlockMesh(grid, dims, self.rank, force_exchange=self._force) if self.blocks else SlabMesh(grid, self.size, self.rank)
        # complex states live as planar (re, im) pairs of the real type on the device (pde_hip/complex_expr.py): the equation is the real
        # system of its parts, and every operand whose ghost layers travel is ONE real component - the exchange is dtype-agnostic at that level,
        # like the reference's `_MPIBC` (pde/backends/numba_mpi/backend.py:30-194) is for its complex arrays
        from .backend import real_dtype_of

        self.is_complex = self.dtype.kind == "c"
        self.info = GridInfo(self.mesh.local_shape, grid.discretization, real_dtype_of(self.dtype))
        exchanging = self.size > 1 or self._force
        self.comm = create_communicator(self.lib, self.control) if exchanging else None
        # the evaluator: HipBackendMixin.make_expression_rhs with this object answering for the box (faces, arrays, layout)
        self.erhs = HipBackendMixin.make_expression_rhs(self, eq, state)
        self.ncomp = int(getattr(self.erhs, "ncomp", 1))
        parts = getattr(self.erhs, "parts", [self.erhs])
        # the exchange as the C loops read it (pdehip_exchange_t: pdehip_jit_euler_run / _rk_run / _euler_adaptive_run exchange the ghost
        # layers of a pass's operand themselves and reduce the adaptive error over the ranks - the reference jits its loops around the MPI
        # calls, pde/solvers/explicit_mpi.py:133-226); the Python-driven steppers (hooks, noise, integrals) call `self.exchange`
        self._exchange_desc = None
        if exchanging:
            d = self._exchange_desc = _abi.Exchange()
            d.comm = self.comm.value if hasattr(self.comm, "value") else self.comm
            d.blocks = int(self.blocks)
            d.lower = 0 if self._force else (-1 if getattr(self.mesh, "lower", None) is None else int(self.mesh.lower))
            d.upper = 0 if self._force else (-1 if getattr(self.mesh, "upper", None) is None else int(self.mesh.upper))
            for i in range(6):
                d.nb6[i] = int(self.mesh.nb6[i]) if self.blocks else -1
        # The C loops find the communicator for the MAX all-reduce of the adaptive error through a pass that carries the exchange
        # descriptor - only passes with operators do.  An equation WITHOUT differential operators on a decomposed grid has none: its
        # adaptive loop then runs from Python with `reduce_error` (each rank would otherwise pick its own step sizes, unlike the
        # reference's `sync_errors`, pde/solvers/base.py:577-590).
        desc = self._exchange_desc if os.environ.get("PDEHIP_DECOMP_LOOPS", "1") != "0" else None
        in_loops = desc is not None and any(tb is not None for part in parts for tb in part.pass_faces)
        if hasattr(self.erhs, "parts"):
            self.erhs.reduces_error_in_loops = in_loops
        for part in parts:
            part._reduce = self._sum_over_ranks if self.size > 1 else None
            part._pass_by_pass = True
            part._two_ok = False
            part._exchange = self.exchange if exchanging else None
            part._exchange_desc = desc
            part.reduces_error_in_loops = in_loops
        # every rank must run the same passes in the same order (the exchanges pair up by issue order): checked, not assumed
        plans = ["\n".join(part.plan.describe()) for part in parts]
        if any(other != plans for other in self.control.allgather(plans)):
            msg = "decomposed stepping: the ranks planned different pass sequences for the same expression"
            raise RuntimeError(msg)
        self._state = None

    # --- what make_expression_rhs asks (see HipBackendMixin._expression_*) ---------------------------------------------------
    def _expression_info(self, grid, dtype):
        return self.info

    def _expression_faces(self, grid, bc, comp, part=None):
        """``part`` "re" / "im": the conditions of the real / imaginary part of a complex operand (``convert_bcs``)."""
        rank = 0 if comp is None else (2 if isinstance(comp, tuple) else 1)
        bcs = grid.get_boundary_conditions(bc, rank=rank)
        kw = {} if comp is None else {"comp_shape": (grid.num_axes,) * rank, "component": comp}
        if part is not None:
            kw["part"] = part
        return self.mesh.block_faces(bcs, **kw) if self.blocks else self.mesh.slab_faces(bcs, force_exchange=self._force, **kw)

    def _expression_aux(self, info, host):
        from .device import DeviceArray

        return DeviceArray(info).set_valid(np.ascontiguousarray(self.mesh.extract(host)), self.stream)

    # --- data plane ---------------------------------------------------------------------------------------------
    def exchange(self, arr) -> None:
        """Fill the ghost layers of ``arr`` (one component) on every face that has a neighbour."""
        if self.blocks:
            self.lib.block_exchange(self.comm, self.info.ref, self.mesh.nb6, arr.ptr, self.stream)
        else:
            lo = 0 if self._force else (-1 if self.mesh.lower is None else int(self.mesh.lower))
            up = 0 if self._force else (-1 if self.mesh.upper is None else int(self.mesh.upper))
            self.lib.halo_exchange(self.comm, self.info.ref, arr.ptr, lo, up, self.stream)

    def _sum_over_ranks(self, value: float) -> float:
        """Integrals over the grid: the partial integrals of the boxes added in rank order (deterministic; equal to the serial sum
        up to rounding, like the reference's MPI all-reduce)."""
        return float(sum(float(v) for v in self.control.allgather(float(value))))

    def _max_over_ranks(self, err_dev) -> None:
        """MAX all-reduce of the device scalar ``err_dev`` (``pdehip_allreduce_max``: RCCL, NaN wins)."""
        self.lib.allreduce_max(self.comm, err_dev.ptr, self.stream)

    def state_array(self):
        from .device import DeviceArray

        if self._state is None:
            if self.is_complex:
                # (re, im) planes per complex component: ncomp counts the REAL components
                lead = (self.ncomp // 2,) if self.ncomp > 2 else ()
                self._state = DeviceArray(self.info, lead + (2,), complex_pairs=True)
            else:
                self._state = DeviceArray(self.info, (self.ncomp,) if self.ncomp > 1 else ())
        return self._state

    def scatter(self, global_valid: np.ndarray):
        return self.state_array().set_valid(np.ascontiguousarray(self.mesh.extract(np.asarray(global_valid)), dtype=self.dtype), self.stream)

    def gather(self, arr, root: int | None = None, out: np.ndarray | None = None):
        """The global valid array (leading component axes first) from the boxes of all ranks, on every rank or on ``root`` only."""
        from .mesh import BlockMesh

        local = arr.get_valid(stream=self.stream)
        nd = len(self.grid.shape)
        lead = local.ndim - nd
        boxes = []
        for r in range(self.size):
            m = BlockMesh(self.grid, self.dims, r)
            boxes.append((slice(None),) * lead + tuple(slice(lo, hi) for lo, hi in zip(m.lo, m.hi)))
        return gather_parts(self.control, local, boxes, tuple(local.shape[:lead]) + tuple(self.grid.shape), root, out)

    # --- steppers -----------------------------------------------------------------------------------------------
    def make_noise_step(self, eq, dt: float):
        """``add_noise(array)`` of an Euler-Maruyama step on the box of this rank - additive Gaussian white noise of constant
        variance ``eq.noise`` (one value, or one per field): ``state += sqrt(dt) * sqrt(noise / cell_volume) * dW`` with dW from the
        device generator (``HipBackendMixin._make_noise_step``); every rank draws from its own part of the counter space, so the
        boxes get independent noise (the reference's MPI nodes each run their own generator).  None for deterministic equations."""
        if not getattr(eq, "is_sde", False):
            return None
        for cls in type(eq).__mro__:
            if "make_noise_variance" in vars(cls) and cls.__name__ not in {"SDEBase", "PDEBase"}:
                msg = "decomposed stepping supports additive noise of constant variance (a custom `make_noise_variance` runs on one device)"
                raise NotImplementedError(msg)
            if "make_noise_variance" in vars(cls):
                break
        if getattr(eq, "use_noise_realization", False) or not getattr(eq, "use_noise_variance", True):
            msg = "decomposed stepping supports Gaussian white noise given by its variance only"
            raise NotImplementedError(msg)
        try:
            noise = np.broadcast_to(np.asarray(getattr(eq, "noise", 0), dtype=float), (self.ncomp,))
        except ValueError:
            noise = None
        if noise is None or (noise < 0).any():
            msg = "decomposed stepping needs one non-negative noise variance per field"
            raise NotImplementedError(msg)
        cell_volume = float(np.prod(self.grid.discretization))
        cells = int(np.prod(self.grid.shape))
        scales = [float(np.sqrt(dt) * np.sqrt(v / cell_volume)) for v in noise]
        rng = getattr(eq, "rng", None)
        seed = int(self.control.broadcast(int((rng if rng is not None else np.random.default_rng()).integers(0, 2**32))))
        counter = [0]
        first = self.rank * self.ncomp * cells      # this rank's part of the counter space (a box never has more cells than the grid)

        def add_noise(arr) -> None:
            flat = arr.flat() if self.ncomp > 1 else None
            for k in range(self.ncomp):
                if scales[k] != 0:
                    ptr = arr.ptr if flat is None else flat.component(k).ptr
                    self.lib.add_gaussian_noise(self.info.ref, 1, ptr, scales[k], seed, counter[0], first + k * cells, self.stream)
            counter[0] += 1

        return add_noise

    def make_stepper(self, scheme: str = "euler", dt: float = 1e-3, *, adaptive: bool = False, tolerance: float = 1e-4, dt_min: float = 1e-10,
                     dt_max: float = 1e10, post_step=None):
        """``stepper(state_array, t_start, t_end) -> (state_array, t_last)`` on the box of this rank, plus its ``info`` dict.
        ``post_step(array, t) -> array``: called after every (accepted) step, after the noise increment."""
        from types import SimpleNamespace

        from .backend import HipBackendMixin

        if scheme not in ("euler", "runge-kutta"):
            msg = f"decomposed stepping supports the schemes euler and runge-kutta (got {scheme})"
            raise NotImplementedError(msg)
        solver = SimpleNamespace(pde=self.eq, adaptive=bool(adaptive), tolerance=float(tolerance), dt_min=float(dt_min), dt_max=float(dt_max),
                                 info={"dt": float(dt), "steps": 0})
        proxy = SimpleNamespace(grid=self.grid, dtype=self.dtype)
        hook = post_step
        add_noise = self.make_noise_step(self.eq, float(dt))
        if add_noise is not None:
            # Euler-Maruyama (pde/solvers/euler.py:66-147): deterministic Euler step, then the noise increment
            if adaptive:
                msg = "Cannot use adaptive stepping with stochastic equation"   # pde/solvers/base.py:446-449
                raise RuntimeError(msg)
            if scheme != "euler":
                msg = "decomposed stepping supports stochastic equations with the Euler scheme"
                raise NotImplementedError(msg)

            def post_step(arr, t):
                add_noise(arr)
                return arr if hook is None else hook(arr, t)

        solver.info["stochastic"] = add_noise is not None
        step = HipBackendMixin._make_expression_stepper(self, solver, proxy, erhs=self.erhs, scheme=scheme, post_step=post_step,
                                                        reduce_error=self._max_over_ranks if self.size > 1 else None)
        return step, solver.info

    def solve(self, global_valid: np.ndarray, t_range: float, dt: float | None, solver: str = "euler", *, tolerance: float = 1e-4,
              dt_min: float = 1e-10, dt_max: float = 1e10) -> tuple[np.ndarray, dict[str, Any]]:
        """Decomposed twin of ``eq.solve(...)`` with ``tracker=None``; returns (global final state, info)."""
        adaptive = dt is None
        step, info = self.make_stepper(solver, 1e-3 if dt is None else dt, adaptive=adaptive, tolerance=tolerance, dt_min=dt_min, dt_max=dt_max)
        arr = self.scatter(global_valid)
        arr, t_last = step(arr, 0.0, float(t_range))
        out = dict(info)
        out.update(world_size=self.size, decomposition=list(self.dims), t_final=t_last)
        return self.gather(arr), out

    def close(self) -> None:
        if self.comm is not None:
            self.lib.stream_synchronize(self.stream)
            self.lib.comm_destroy(self.comm)
            self.comm = None
        if getattr(self, "stream", None) is not None:
            stream, self.stream = self.stream, None
            try:
                self.lib.stream_synchronize(stream)
                self.lib.stream_destroy(stream)
            except Exception:  # noqa: BLE001 - releasing a resource at the end of a run must not turn a finished run into a failure
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass
