"""Slab-parallel explicit steppers: one process per GPU, halo exchange over RCCL (xGMI).

Replaces the reference's MPI path — ``ExplicitMPISolver`` (``pde/solvers/explicit_mpi.py:133-226``),
``_MPIBC`` ghost exchange (``pde/grids/boundaries/local.py:561-662``,
``pde/backends/numba_mpi/backend.py:30-194``) and the MAX all-reduce of the adaptive error
(``pde/backends/base.py:678-712``) — with ``torch.distributed`` point-to-point ops (backend
``nccl`` == RCCL on ROCm).  The reference exchanges faces with *blocking* sends inside every
right-hand side; here the exchange of step s+1 overlaps the interior kernel of step s:

    comp stream : ghosts(y/z faces) ─ interior kernel (layers 2..n-1) ───────────┐
    halo stream : wait(recv of cur) ─ boundary kernels (layers 1, n) ─ send/recv of nxt faces

Axis-0 slabs keep every face contiguous (no pack kernel).  The numerical kernels are reached
through an *engine* object: :class:`HipEngine` (product: libpdehip + HIP streams) — the test-suite
injects a CPU engine built on the oracle to exercise this orchestration with ``gloo`` on world
size 2.  The engine is never chosen implicitly: without a GPU ``HipEngine`` raises.
"""

from __future__ import annotations

import ctypes as C
from typing import Any

import numpy as np

from . import _abi
from .backend import RhsSpec, convert_bcs
from .mesh import SlabMesh


# ---------------------------------------------------------------------------------------------
# engine: where kernels run
# ---------------------------------------------------------------------------------------------
class HipEngine:
    """libpdehip on the current HIP device; memory and streams come from torch (plumbing)."""

    device_type = "cuda"

    def __init__(self, device: int | None = None):
        import torch

        from ._lib import require_device

        self.torch = torch
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.lib = require_device(device)
        torch.cuda.set_device(device)
        self.device = torch.device("cuda", device)
        self.comp = torch.cuda.Stream(device=self.device)
        self.halo = torch.cuda.Stream(device=self.device)

    # native RCCL communicator -----------------------------------------------------------------
    def make_comm(self, dist, group, size: int, rank: int):
        """Create libpdehip's own RCCL communicator (ncclUniqueId distributed through ``dist``).

        libpdehip resolves RCCL from the librccl.so torch already loaded, so there is one RCCL in
        the process.  Returns ``None`` when ``PDEHIP_COMM=torch`` asks for the torch P2P path.
        """
        import os

        if os.environ.get("PDEHIP_COMM", "native") == "torch":
            return None
        path = os.path.join(os.path.dirname(self.torch.__file__), "lib", "librccl.so")
        if not os.path.exists(path):
            path = "librccl.so"
        uid = C.create_string_buffer(128)
        if rank == 0:
            self.lib.comm_unique_id(path.encode(), uid)
        if size > 1:
            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = C.create_string_buffer(box[0], 128)
        comm = C.c_void_p()
        self.lib.comm_create(path.encode(), uid, rank, size, C.byref(comm))
        return comm

    # layout / memory -------------------------------------------------------------------------
    def layout(self, g: _abi.Grid) -> dict[str, int]:
        lay = (C.c_int64 * 8)()
        self.lib.layout(C.byref(g), lay)
        return {"comp_elems": int(lay[2]), "slack": int(lay[6]), "layer_pitch": int(lay[7])}

    def alloc(self, nelems: int, dtype) -> Any:
        tdt = self.torch.float64 if np.dtype(dtype) == np.float64 else self.torch.float32
        return self.torch.zeros(nelems, dtype=tdt, device=self.device)

    def upload_f64(self, arr: np.ndarray):
        t = self.torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).to(self.device)
        t.ptr = t.data_ptr()
        return t

    def set_valid(self, g, buf, host_valid: np.ndarray) -> None:
        stage = self.torch.from_numpy(np.ascontiguousarray(host_valid)).to(self.device)
        self.lib.valid_to_full(C.byref(g), 1, stage.data_ptr(), buf.data_ptr(), self.stream_ptr(self.comp))
        self.comp.synchronize()

    def get_valid(self, g, buf, shape, dtype) -> np.ndarray:
        tdt = self.torch.float64 if np.dtype(dtype) == np.float64 else self.torch.float32
        stage = self.torch.empty(int(np.prod(shape)), dtype=tdt, device=self.device)
        self.lib.full_to_valid(C.byref(g), 1, buf.data_ptr(), stage.data_ptr(), self.stream_ptr(self.comp))
        self.comp.synchronize()
        return stage.cpu().numpy().reshape(shape)

    # kernels ---------------------------------------------------------------------------------------
    def stream_ptr(self, stream) -> int | None:
        return stream.cuda_stream if stream is not None else None

    def call(self, name: str, stream, *args) -> None:
        getattr(self.lib, name)(*args, self.stream_ptr(stream))

    # stream plumbing --------------------------------------------------------------------------------
    def use(self, stream):
        return self.torch.cuda.stream(stream)

    def record(self, stream):
        ev = self.torch.cuda.Event()
        ev.record(stream)
        return ev

    def wait(self, stream, event) -> None:
        if event is not None:
            stream.wait_event(event)

    def synchronize(self) -> None:
        self.comp.synchronize()
        self.halo.synchronize()

    def scalar(self):
        return self.torch.zeros(1, dtype=self.torch.float64, device=self.device)


class _NoUpload:
    """Face-table `upload` stub for checks that only look at kinds / indices (array-valued faces are rejected anyway)."""

    ptr = 0

    def __init__(self, arr):
        pass


# ---------------------------------------------------------------------------------------------
# the slab stepper
# ---------------------------------------------------------------------------------------------
class SlabStepper:
    """Explicit Euler / RK4 / RKF45 for Diffusion and Cahn–Hilliard on an axis-0 slab decomposition."""

    def __init__(self, eq, grid, dtype=np.float64, *, engine=None, group=None, force_exchange: bool = False):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.engine = engine if engine is not None else HipEngine()
        self.eq, self.grid, self.dtype = eq, grid, np.dtype(dtype)
        self.mesh = SlabMesh(grid, self.size, self.rank)
        sub = self.mesh.subgrid
        self.n = self.mesh.n_local
        self.g = _abi.make_grid(sub.shape, sub.discretization, self.dtype)
        lay = self.engine.layout(self.g)
        self.layer_pitch, self.comp_elems = lay["layer_pitch"], lay["comp_elems"]
        self.nelems = self.comp_elems + lay["slack"]
        self.itemsize = self.dtype.itemsize
        # neighbours; world size 1 + periodic axis 0 can be forced through the exchange path
        self.lower, self.upper = self.mesh.lower, self.mesh.upper
        skip = set(self.mesh.exchanged_faces)
        if force_exchange and self.size == 1 and grid.periodic[0]:
            self.lower = self.upper = 0
            skip = {(0, False), (0, True)}
        self.exchanging = self.lower is not None or self.upper is not None
        # right-hand side description
        name = eq.__class__.__name__
        if name == "DiffusionPDE":
            self.kind, self.param = _abi.RHS_DIFFUSION, float(eq.diffusivity)
            bc_c = bc_mu = grid.get_boundary_conditions(eq.bc, rank=0)
        elif name == "CahnHilliardPDE":
            self.kind, self.param = _abi.RHS_CAHN_HILLIARD, float(eq.interface_width)
            bc_c = grid.get_boundary_conditions(eq.bc_c, rank=0)
            bc_mu = grid.get_boundary_conditions(eq.bc_mu, rank=0)
        elif name == "PDE":
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
            self.kind, self.param = match
            # one condition per operator name, inner and outer laplace alike (pde/pdes/pde.py:329-343)
            bc_c = bc_mu = grid.get_boundary_conditions(pde_bc_for(eq, var, "laplace"), rank=0)
        else:
            msg = f"slab stepper has no fused right-hand side for {name}"
            raise NotImplementedError(msg)
        self.faces_c = convert_bcs(self.mesh.sub_boundaries(bc_c), skip=skip, upload=self.engine.upload_f64)
        self.faces_mu = convert_bcs(self.mesh.sub_boundaries(bc_mu), skip=skip, upload=self.engine.upload_f64)
        self._bufs: dict[str, Any] = {}
        self.err = self.engine.scalar()
        self.steps_done = 0
        # libpdehip's own RCCL communicator (product path); engines without one (the CPU test
        # engine, or PDEHIP_COMM=torch) use torch.distributed point-to-point ops instead
        self.comm = None
        if self.exchanging or self.size > 1:
            make = getattr(self.engine, "make_comm", None)
            if make is not None:
                try:
                    self.comm = make(dist, group, self.size, self.rank)
                    ok = self.comm is not None
                except (RuntimeError, OSError) as err:  # e.g. RCCL library not resolvable
                    import logging

                    logging.getLogger("pde_hip.distributed").warning("native RCCL communicator unavailable (%s); using torch P2P ops", err)
                    self.comm, ok = None, False
                if self.size > 1:
                    # all ranks must agree on the transport, otherwise their collectives would not match
                    flag = self.engine.torch.tensor([1 if ok else 0], device=self.engine.device)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                    if int(flag.item()) == 0 and self.comm is not None:
                        self.engine.lib.comm_destroy(self.comm)
                        self.comm = None
        # the C ABI wants the full face table (physical faces on, exchanged ones are skipped there)
        self._rhs_c = None
        if self.comm is not None and self.kind == _abi.RHS_DIFFUSION:
            self._faces_all = convert_bcs(self.mesh.sub_boundaries(bc_c), skip=skip, upload=self.engine.upload_f64)
            self._rhs_c = _abi.RHS()
            self._rhs_c.kind = _abi.RHS_DIFFUSION
            self._rhs_c.param = self.param
            self._faces_all.copy_into(self._rhs_c.bc_c)
        # two steps per sweep (temporal blocking, two halo layers exchanged every other step): decided from GLOBAL
        # information only, so that all ranks take the same path
        self._euler2 = False
        x_ok = bool(grid.periodic[0])
        if not x_ok:
            # non-periodic slowest axis: the first / last rank apply the physical face inside the kernel, which needs a
            # local first-order face with scalar coefficients on BOTH ends of the GLOBAL grid (checked on every rank)
            glob = convert_bcs(bc_c, upload=_NoUpload).c
            x_ok = all(glob[s].kind == _abi.BC_ORDER1 and glob[s].flags == 0 and glob[s].index1 == (grid.shape[0] - 1 if s else 0)
                       for s in (0, 1))
        if self._rhs_c is not None and self.exchanging and x_ok and min(self.mesh.counts) >= 4 and grid.num_axes == 3:
            ok = C.c_int(0)
            self.engine.lib.slab_euler2_supported(C.byref(self.g), C.byref(self._rhs_c), C.byref(ok))
            self._euler2 = bool(ok.value)
        # Cahn-Hilliard: the right-hand side in ONE sweep (mu in registers) after ONE exchange of two layers of c
        self._ch_rhs = None
        x_ok_ch = bool(grid.periodic[0])
        if not x_ok_ch and self.kind == _abi.RHS_CAHN_HILLIARD:   # both fields need local scalar faces on the global ends
            tabs = [convert_bcs(b, upload=_NoUpload).c for b in (bc_c, bc_mu)]
            x_ok_ch = all(t[s].kind == _abi.BC_ORDER1 and t[s].flags == 0 and t[s].index1 == (grid.shape[0] - 1 if s else 0)
                          for t in tabs for s in (0, 1))
        if (self.comm is not None and self.kind == _abi.RHS_CAHN_HILLIARD and self.exchanging and x_ok_ch
                and min(self.mesh.counts) >= 2 and grid.num_axes == 3):
            rhs = _abi.RHS()
            rhs.kind, rhs.param = _abi.RHS_CAHN_HILLIARD, self.param
            self.faces_c.copy_into(rhs.bc_c)
            self.faces_mu.copy_into(rhs.bc_mu)
            ok = C.c_int(0)
            self.engine.lib.slab_ch_supported(C.byref(self.g), C.byref(rhs), C.byref(ok))
            if ok.value:
                self._ch_rhs = rhs

    # --- buffers ---------------------------------------------------------------------------------
    def buf(self, name: str):
        """Slab array (n + 2 layers) as a flat tensor; the allocation holds ONE MORE layer on either side, so the same
        memory is also a slab array with two halo layers per side (``ext_ptr``) for the two-level kernels."""
        if name not in self._bufs:
            full = self.engine.alloc(self.nelems + 2 * self.layer_pitch, self.dtype)
            self._bufs[name] = full[self.layer_pitch : self.layer_pitch + self.nelems]
        return self._bufs[name]

    def ext_ptr(self, buf) -> int:
        """Pointer to the array with two halo layers per side that contains ``buf`` (own layers 2..n+1)."""
        return buf.data_ptr() - self.layer_pitch * self.itemsize

    def layer(self, buf, index: int):
        """Full layer ``index`` (0 = lower ghost layer) as a flat tensor view (contiguous)."""
        return buf[index * self.layer_pitch : (index + 1) * self.layer_pitch]

    def _sub(self, first_layer: int, count: int) -> tuple[_abi.Grid, int]:
        """Grid descriptor + element offset of the sub-slab of valid layers [first, first+count)."""
        shape = (count, *self.mesh.subgrid.shape[1:])
        g = _abi.make_grid(shape, self.mesh.subgrid.discretization, self.dtype)
        return g, (first_layer - 1) * self.layer_pitch

    def ptr(self, buf, elem_offset: int = 0) -> int:
        return buf.data_ptr() + elem_offset * self.itemsize

    # --- halo exchange ----------------------------------------------------------------------------
    def start_exchange(self, buf, stream):
        """Post send/recv of the two boundary layers of ``buf`` on ``stream``; returns when enqueued.

        Order per peer: the "downward" pair first, then the "upward" pair, so that with RCCL
        (which matches sends and receives to one peer in issue order, tags are ignored) the 2-rank
        periodic ring and the 1-rank self exchange pair up correctly.
        """
        if not self.exchanging:
            return
        if self.comm is not None:   # native path: ncclSend/ncclRecv group issued by libpdehip
            lower = -1 if self.lower is None else self.lower
            upper = -1 if self.upper is None else self.upper
            self.engine.lib.halo_exchange(self.comm, C.byref(self.g), self.ptr(buf), lower, upper, self.engine.stream_ptr(stream))
            return
        dist = self.dist
        n = self.n
        ops = []
        if self.lower is not None:
            ops.append(dist.P2POp(dist.isend, self.layer(buf, 1), self.lower, self.group, 0))
        if self.upper is not None:
            ops.append(dist.P2POp(dist.irecv, self.layer(buf, n + 1), self.upper, self.group, 0))
            ops.append(dist.P2POp(dist.isend, self.layer(buf, n), self.upper, self.group, 1))
        if self.lower is not None:
            ops.append(dist.P2POp(dist.irecv, self.layer(buf, 0), self.lower, self.group, 1))
        with self.engine.use(stream):
            works = dist.batch_isend_irecv(ops)
            for w in works:
                w.wait()  # device-side wait on `stream` for NCCL; host wait for gloo

    # --- building blocks ---------------------------------------------------------------------------
    def _ghosts(self, faces, buf, stream) -> None:
        self.engine.call("set_ghost_cells", stream, C.byref(self.g), 1, faces.c, self.ptr(buf))

    def _lap(self, mode: str, stream, first: int, count: int, src, dst, *, y=None, s1=1.0, s2=1.0) -> None:
        if count <= 0:
            return
        g, off = self._sub(first, count)
        if mode == "euler":
            self.engine.call("laplace_euler", stream, C.byref(g), self.ptr(src, off), self.ptr(y, off), self.ptr(dst, off), s1, s2)
        elif mode == "scaled":
            self.engine.call("laplace_scaled", stream, C.byref(g), self.ptr(src, off), self.ptr(dst, off), s1, s2)
        else:
            self.engine.call("cahn_hilliard_mu", stream, C.byref(g), self.ptr(src, off), self.ptr(dst, off), s1)

    def _stencil_pass(self, mode, faces, src, dst, **kw) -> None:
        """ghosts + exchange + stencil over the whole slab on the comp stream (no overlap)."""
        comp = self.engine.comp
        self.start_exchange(src, comp)
        self._ghosts(faces, src, comp)
        self._lap(mode, comp, 1, self.n, src, dst, **kw)

    def rhs_scaled(self, y, k_out, dt: float) -> None:
        """k_out = dt * rhs(y)  (same sequence as pdehip_rhs_scaled, plus the halo exchange)."""
        if self.kind == _abi.RHS_DIFFUSION:
            self._stencil_pass("scaled", self.faces_c, y, k_out, s1=self.param, s2=dt)
        elif self._ch_rhs is not None:
            self._ch_sweep(y, k_out, dt, euler=False)
        else:
            mu = self.buf("mu")
            self._stencil_pass("mu", self.faces_c, y, mu, s1=self.param)
            self._stencil_pass("scaled", self.faces_mu, mu, k_out, s1=1.0, s2=dt)

    def _ch_sweep(self, c, out, dt: float, *, euler: bool) -> None:
        """Exchange two layers of c per side, then the fused Cahn-Hilliard sweep (mu in registers) on the comp stream."""
        lower = -1 if self.lower is None else self.lower
        upper = -1 if self.upper is None else self.upper
        self.engine.lib.slab_ch_sweep(self.comm, C.byref(self.g), C.byref(self._ch_rhs), lower, upper, self.ext_ptr(c),
                                      self.ext_ptr(out), dt, 1 if euler else 0, self.engine.stream_ptr(self.engine.comp))

    def lincomb(self, out, y, coefs, ks) -> None:
        cf = (C.c_double * len(coefs))(*coefs)
        kp = (C.c_void_p * len(ks))(*[self.ptr(k) for k in ks])
        self.engine.call("lincomb", self.engine.comp, C.byref(self.g), 1, self.ptr(out), self.ptr(y), len(ks), cf, kp)

    # --- Euler ----------------------------------------------------------------------------------------
    def euler_steps(self, cur, nxt, dt: float, nsteps: int):
        """``nsteps`` Euler steps ping-ponging cur/nxt; returns the buffer holding the result."""
        eng, n = self.engine, self.n
        comp, halo = eng.comp, eng.halo
        if self.kind != _abi.RHS_DIFFUSION or not self.exchanging:
            for _ in range(nsteps):
                if self._ch_rhs is not None:
                    self._ch_sweep(cur, nxt, dt, euler=True)
                    cur, nxt = nxt, cur
                    continue
                if self.kind == _abi.RHS_DIFFUSION:
                    self._stencil_pass("euler", self.faces_c, cur, nxt, y=cur, s1=self.param, s2=dt)
                else:
                    mu = self.buf("mu")
                    self._stencil_pass("mu", self.faces_c, cur, mu, s1=self.param)
                    self._stencil_pass("euler", self.faces_mu, mu, nxt, y=cur, s1=1.0, s2=dt)
                cur, nxt = nxt, cur
            return cur
        if self._rhs_c is not None:
            # native overlapped loop: the whole step sequence is enqueued by ONE C call
            lower = -1 if self.lower is None else self.lower
            upper = -1 if self.upper is None else self.upper
            res = C.c_void_p()
            run = eng.lib.slab_euler2_run if self._euler2 and nsteps >= 2 else eng.lib.slab_euler_run
            run(self.comm, C.byref(self.g), C.byref(self._rhs_c), lower, upper, self.ptr(cur), self.ptr(nxt),
                dt, nsteps, C.byref(res), eng.stream_ptr(comp))
            return cur if res.value == self.ptr(cur) else nxt
        # overlapped diffusion path (torch P2P / CPU test engine) ------------------------------------
        eng.wait(halo, eng.record(comp))
        self.start_exchange(cur, halo)            # ghost layers of the initial state
        ev_boundary = None
        for _ in range(nsteps):
            eng.wait(comp, ev_boundary)           # boundary layers of `cur` were written on halo stream
            self._ghosts(self.faces_c, cur, comp) # y/z faces + physical x faces
            ev_ghosts = eng.record(comp)
            # interior layers need no exchanged data
            self._lap("euler", comp, 2, n - 2, cur, nxt, y=cur, s1=self.param, s2=dt)
            # boundary layers: need the received ghost layers (halo stream order) and ev_ghosts
            eng.wait(halo, ev_ghosts)
            self._lap("euler", halo, 1, 1, cur, nxt, y=cur, s1=self.param, s2=dt)
            if n > 1:
                self._lap("euler", halo, n, 1, cur, nxt, y=cur, s1=self.param, s2=dt)
            ev_boundary = eng.record(halo)
            self.start_exchange(nxt, halo)        # overlaps the interior kernel on comp (and the next ghosts)
            # the next iteration overwrites `cur` (as its `nxt`): interior of this step must be done
            eng.wait(halo, eng.record(comp))
            cur, nxt = nxt, cur
        eng.wait(comp, ev_boundary)
        return cur

    # --- Runge–Kutta ---------------------------------------------------------------------------------
    def rk4_step(self, y, dt: float) -> None:
        k1, k2, k3, k4, tmp = (self.buf(n) for n in ("k1", "k2", "k3", "k4", "tmp"))
        self.rhs_scaled(y, k1, dt)
        self.lincomb(tmp, y, [0.5], [k1])
        self.rhs_scaled(tmp, k2, dt)
        self.lincomb(tmp, y, [0.5], [k2])
        self.rhs_scaled(tmp, k3, dt)
        self.lincomb(tmp, y, [1.0], [k3])
        self.rhs_scaled(tmp, k4, dt)
        self.engine.call("rk4_combine", self.engine.comp, C.byref(self.g), 1, self.ptr(y), self.ptr(k1), self.ptr(k2), self.ptr(k3), self.ptr(k4))

    _B = [[1 / 4], [3 / 32, 9 / 32], [1932 / 2197, -7200 / 2197, 7296 / 2197], [439 / 216, -8.0, 3680 / 513, -845 / 4104],
          [-8 / 27, 2.0, -3544 / 2565, 1859 / 4104, -11 / 40]]

    def rkf45_attempt(self, y, ynew, dt: float) -> float:
        """One RKF45 attempt; returns the error already MAX-reduced over all ranks."""
        ks = [self.buf(f"k{i}") for i in range(1, 7)]
        tmp = self.buf("tmp")
        self.rhs_scaled(y, ks[0], dt)
        for s, b in enumerate(self._B):
            self.lincomb(tmp, y, b, ks[: s + 1])
            self.rhs_scaled(tmp, ks[s + 1], dt)
        kp = (C.c_void_p * 6)(*[self.ptr(k) for k in ks])
        self.engine.call("rkf45_combine", self.engine.comp, C.byref(self.g), 1, self.ptr(y), self.ptr(ynew), kp, self.err.data_ptr())
        return self.sync_max(self.err)

    def sync_max(self, err_tensor) -> float:
        """MAX all-reduce of the error scalar (``make_mpi_synchronizer``, backends/base.py:678-712)."""
        if self.size > 1 and self.comm is not None:
            self.engine.lib.allreduce_max(self.comm, err_tensor.data_ptr(), self.engine.stream_ptr(self.engine.comp))
            self.engine.synchronize()
            return float(err_tensor.cpu()[0])
        if self.size > 1:
            with self.engine.use(self.engine.comp):
                # NaN must win the reduction like np.max: reduce a NaN flag alongside
                flag = err_tensor.isnan().to(err_tensor.dtype)
                val = err_tensor.nan_to_num(nan=0.0)
                both = self.engine.torch.cat([val, flag])
                self.dist.all_reduce(both, op=self.dist.ReduceOp.MAX, group=self.group)
            self.engine.synchronize()
            v, f = (float(x) for x in both.cpu())
            return float("nan") if f > 0 else v
        self.engine.synchronize()
        return float(err_tensor.cpu()[0])

    # --- user level --------------------------------------------------------------------------------------
    def scatter(self, global_valid: np.ndarray):
        """Upload this rank's block of a (replicated) global initial state; returns the buffer."""
        buf = self.buf("state_a")
        self.engine.set_valid(self.g, buf, self.mesh.extract(global_valid).astype(self.dtype))
        return buf

    def gather_local(self, buf) -> np.ndarray:
        return self.engine.get_valid(self.g, buf, self.mesh.subgrid.shape, self.dtype)

    def gather(self, buf) -> np.ndarray:
        """All ranks receive the global valid array (only for tests / tracker interrupts)."""
        local = self.gather_local(buf)
        if self.size == 1:
            return local
        blocks: list[Any] = [None] * self.size
        self.dist.all_gather_object(blocks, local, group=self.group)
        return np.concatenate(blocks, axis=0)

    def solve(self, global_valid: np.ndarray, t_range: float, dt: float | None, solver: str = "euler", *, tolerance: float = 1e-4,
              dt_min: float = 1e-10, dt_max: float = 1e10) -> tuple[np.ndarray, dict[str, Any]]:
        """Slab-parallel twin of ``eq.solve(...)`` with ``tracker=None``; returns (global final state, info)."""
        from .solvers import OnlineStatistics, make_dt_adjuster

        cur = self.scatter(global_valid)
        nxt = self.buf("state_b")
        info: dict[str, Any] = {"steps": 0, "world_size": self.size}
        if dt is not None:
            steps = max(1, round(t_range / dt))
            if solver == "euler":
                cur = self.euler_steps(cur, nxt, dt, steps)
            elif solver == "runge-kutta":
                for _ in range(steps):
                    self.rk4_step(cur, dt)
            else:
                msg = f"slab stepper does not support solver {solver}"
                raise NotImplementedError(msg)
            info.update(steps=steps, dt=dt, t_final=(steps - 1) * dt + dt)
        else:
            if solver != "runge-kutta":
                msg = "adaptive slab stepping is implemented for runge-kutta (RKF45)"
                raise NotImplementedError(msg)
            adjust = make_dt_adjuster(dt_min, dt_max)
            stats = OnlineStatistics()
            dt_opt, t, steps = 1e-3, 0.0, 0
            while True:
                dt_step = max(min(dt_opt, t_range - t), dt_min)
                error_rel = self.rkf45_attempt(cur, nxt, dt_step) / tolerance
                if error_rel <= 1:
                    steps += 1
                    t += dt_step
                    cur, nxt = nxt, cur
                    stats.add(dt_step)
                if t < t_range:
                    dt_opt = adjust(dt_step, error_rel)
                else:
                    break
            info.update(steps=steps, dt=dt_opt, t_final=t, dt_statistics=stats.to_dict())
        self.engine.synchronize()
        return self.gather(cur), info
