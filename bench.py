"""Benchmark of the hot path: 3-D DiffusionPDE 512^3 fp64, explicit Euler, on N MI355X GPUs.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the
driver launches one rank per GPU through torch.distributed.run - started WITHOUT a launcher, `--gpus N`
spawns its N ranks itself (`spawn_ranks`), and a world size that differs from `--gpus` is an error, never a line.  One "step" = one explicit Euler
step of the whole 512^3 grid (BCs on the fly + fused Laplacian + D*, dt*, +=; two steps per kernel sweep), state
resident in HBM (ping-pong buffers), i.e. 16 algorithmic bytes per cell-step.
N > 1 slab-decomposes the SAME grid along axis 0 (strong scaling) with RCCL halo exchange
overlapped with the interior kernel (pde_hip/distributed.py).

Rank 0 prints ONE JSON line with the whole-job throughput in Mcells/s (cell-steps per second),
plus `roofline` for the dominant kernel (HIP-event timing on the launch stream; `kernel` = the instance name the
library reports for what it dispatched, `traffic` = PMC bytes per launch of exactly that instance from profiles/traffic.json) and, at N = 1,
`cpu_baseline` = the CPU oracle (plain-C restatement of the reference formulas, OpenMP) timed on
this box's host cores on a bounded sample of the same workload (two builds: the exact one and numba's fastmath flag set),
`extra.slab_share_to_self` = the N-GPU slab step of this grid measured on this one GPU with the halo exchange sent to self.
The N > 1 line carries `rccl`: what RCCL reports about its communicator (ranks, devices, PCI bus ids).
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
for _p in (ROOT, ROOT / "py-pde_amd"):
    if str(_p) not in sys.path:
        sys.path.insert(0, str(_p))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s
BYTES_PER_CELL_STEP = 16  # fp64: 1 read + 1 write per cell


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=512, help="cells per axis (BASELINE config: 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-distributed", action="store_true",
                    help="run the slab/RCCL path even on one rank (periodic halo sent to self) - overhead probe")
    ap.add_argument("--decomposition", default="slab",
                    help="N > 1: `slab` (axis-0 slabs, two steps per sweep: the default), `auto` (the reference's rule, pde/grids/_mesh.py:59-93: "
                         "2 x 2 x 2 for 512^3 on 8 ranks) or blocks per axis, e.g. `2,4,1` (decompositions that do not cut the fastest axis run the "
                         "fast block loop, csrc/pdehip_block2_loops.h; the others the exact one-step loop)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-extra", action="store_true", help="skip the operator roofline, the cfg2/cfg3/cfg5 timings and the parity bit")
    ap.add_argument("--no-configs", action="store_true", help="skip only the cfg2/cfg3/cfg5 solves and the 2-D kernel timings (test aid: they take minutes on the host shim)")
    ap.add_argument("--configs-scale", type=int, default=1, help="test aid: the cfg2/cfg3/cfg5 solves on grids this many times smaller per axis (the host shim "
                                                                 "runs them in seconds then; the 2-D kernel timings are skipped); 1 = the BASELINE sizes")
    ap.add_argument("--repeats", type=int, default=9,
                    help="the timed region (EXACTLY --steps steps between two synchronisations) is run this many times; value / ms_per_step "
                         "are the MEDIAN, the line carries every repetition and the minimum (SURVEY.md 8d: min / median of >= 5)")
    return ap.parse_args()


def _stats(samples_ms: list[float]) -> dict:
    """min / median of the repetitions of a timed region (milliseconds per unit), every sample kept."""
    s = sorted(samples_ms)
    med = s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])
    return {"n": len(s), "min": round(s[0], 5), "median": round(med, 5), "max": round(s[-1], 5), "samples": [round(v, 5) for v in samples_ms]}


def cpu_baseline(n: int, seconds: float) -> dict:
    """Time the oracle's Euler loop (reference formulas in C, OpenMP over axis 0 like numba's prange) - twice: the conservative build that
    is the parity checker (-ffp-contract=off) and the same source with the flag set numba's default `fastmath` stands for
    (pde/backends/numba/utils.py:330-336; oracle/Makefile: libpde_oracle_fastmath.so - timing only).  `value` is the faster of the two:
    the fastest honest stand-in for the reference's numba path on these cores; both samples are listed."""
    from oracle import pde_oracle as O
    from pde_hip import _abi

    n_cpu = min(n, 512)
    # threads: what this process may actually use (affinity mask and cgroup CPU quota), not the host's core count —
    # on the GPU boxes 256 logical CPUs are visible but the container gets 16: 256 OpenMP threads run 100x slower
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except OSError:
        pass
    g = _abi.make_grid((n_cpu,) * 3, (1.0,) * 3, np.float64)
    faces = _abi.FaceArray()
    for ax in range(3):
        for side, idx in ((0, n_cpu - 1), (1, 0)):  # periodic: lower reads N-1, upper reads 0
            f = faces[2 * ax + side]
            f.kind, f.flags, f.index1, f.const_v, f.factor1 = _abi.BC_ORDER1, 0, idx, 0.0, 1.0
    rhs = O.make_rhs(_abi.RHS_DIFFUSION, 1.0, faces)
    rng = np.random.default_rng(0)
    a = O.valid_to_full((n_cpu,) * 3, rng.random((n_cpu,) * 3))
    b = np.zeros_like(a)
    res = C.c_void_p()
    builds = [("gcc -O3 -mavx2 -ffp-contract=off -fno-fast-math (the parity checker: no FMA contraction, no reassociation)", O.lib())]
    fast = ROOT / "oracle" / "libpde_oracle_fastmath.so"
    if fast.exists():
        builds.append(("gcc -O3 -mavx2 -mfma -ffp-contract=fast -fassociative-math -fno-signed-zeros -fno-trapping-math -freciprocal-math "
                       "(numba's default fastmath flag set: nsz, arcp, contract, reassoc - pde/backends/numba/utils.py:330-336)", C.CDLL(str(fast))))
    samples = []
    for flags, lib in builds:
        run = lib.oracle_euler_run
        run.restype = C.c_int
        args = (C.byref(g), C.byref(rhs), C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_double(0.1))
        run(*args, C.c_int64(1), C.byref(res))  # warm-up (page faults)
        steps, t0 = 0, time.perf_counter()
        while True:
            run(*args, C.c_int64(2), C.byref(res))
            steps += 2
            el = time.perf_counter() - t0
            if el >= seconds / len(builds) or steps >= 2000:
                break
        samples.append({"value": round(n_cpu**3 * steps / el / 1e6, 2), "unit": "Mcells/s", "kind": "port", "flags": flags,
                        "sample": f"{steps} Euler steps of DiffusionPDE {n_cpu}^3 fp64 periodic (oracle/pde_oracle.c, OpenMP {cores} threads, {el:.1f} s)"})
    best = max(samples, key=lambda smp: smp["value"])
    return {"value": best["value"], "unit": "Mcells/s", "cores": cores, "kind": "port", "sample": best["sample"], "flags": best["flags"],
            "samples": samples, "note": "numba itself is not installable here; both samples are the C restatement of the reference's formulas "
                                        "(oracle/), the faster one is quoted"}


def bench_single(args) -> dict:
    import pde_hip
    from pde_hip.device import DeviceArray

    t_start = time.perf_counter()
    backend = pde_hip.get_backend("hip")
    lib = backend._lib
    n = args.size
    grid = pde_hip.UnitGrid([n, n, n], periodic=True)
    eq = pde_hip.DiffusionPDE(1.0)
    rng = np.random.default_rng(0)
    state = pde_hip.ScalarField.random_uniform(grid, rng=rng)
    spec = backend.make_rhs_spec(eq, state)
    info = spec.info
    a, b = DeviceArray(info).set_valid(state.data), DeviceArray(info)
    stream = C.c_void_p()
    lib.stream_create(C.byref(stream))
    ev = [C.c_void_p() for _ in range(4)]
    for e in ev:
        lib.event_create(C.byref(e))
    res = C.c_void_p()
    dt = 0.1
    cur, nxt = a.ptr, b.ptr
    # dominant kernel alone, HIP events on its launch stream: the two-steps-per-sweep kernel where it covers the
    # grid (temporal blocking: ONE launch = TWO Euler steps, intermediate level in registers), else the one-step kernel
    reps = max(20, min(args.steps, 200))
    done = C.c_int(0)
    lib.diffusion_euler2(info.ref, spec.bc_c.c, cur, nxt, 1.0, dt, C.byref(done), stream)
    steps_per_launch = 2 if done.value else 1
    lib.stream_synchronize(stream)
    kernel_instance = lib.last_kernel_name().decode(errors="replace")   # what launch_euler2 / launch_laplace actually dispatched
    kernel_ms = []
    for _ in range(max(1, args.repeats)):
        lib.event_record(ev[2], stream)
        for _ in range(reps):
            if done.value:
                lib.diffusion_euler2(info.ref, spec.bc_c.c, cur, nxt, 1.0, dt, C.byref(done), stream)
            else:
                lib.laplace_euler(info.ref, cur, cur, nxt, 1.0, dt, stream)
        lib.event_record(ev[3], stream)
        lib.stream_synchronize(stream)
        ms_kernel = C.c_float()
        lib.event_elapsed_ms(ev[2], ev[3], C.byref(ms_kernel))
        kernel_ms.append(ms_kernel.value / reps)
    # (the kernel timing above runs BEFORE warm-up and timed region: with the driver's 20-step regions of 5 ms each the first repetitions
    # used to see the clocks still ramping - r04: 0.2588 ... 0.2434 ms per step over the nine repetitions)
    # W untimed warm-up steps, then ...
    lib.euler_run(info.ref, spec.ref, cur, nxt, dt, args.warmup, C.byref(res), stream)
    lib.stream_synchronize(stream)
    if res.value != cur:
        cur, nxt = nxt, cur
    # timed region: exactly K steps, bracketed by synchronisation - repeated `--repeats` times (each repetition is a complete timed
    # region of its own and continues from the state the previous one left; the line reports the MEDIAN, every sample and the minimum)
    walls, ms_events = [], C.c_float()
    for _ in range(max(1, args.repeats)):
        lib.stream_synchronize(stream)
        t0 = time.perf_counter()
        lib.event_record(ev[0], stream)
        lib.euler_run(info.ref, spec.ref, cur, nxt, dt, args.steps, C.byref(res), stream)
        lib.event_record(ev[1], stream)
        lib.stream_synchronize(stream)
        walls.append(time.perf_counter() - t0)
        lib.event_elapsed_ms(ev[0], ev[1], C.byref(ms_events))
        if res.value != cur:
            cur, nxt = nxt, cur
    wall_stats = _stats([w / args.steps * 1e3 for w in walls])
    wall = wall_stats["median"] * 1e-3 * args.steps
    kernel_stats = _stats(kernel_ms)
    t_kernel = kernel_stats["median"] * 1e-3
    # device-copy ceiling of THIS GPU on the same bytes: hipMemcpyDtoD of the state = 1 read + 1 write per cell
    nbytes_valid = n**3 * 8
    lib.memcpy_d2d(nxt, cur, nbytes_valid, stream)
    lib.stream_synchronize(stream)
    lib.event_record(ev[2], stream)
    for _ in range(10):
        lib.memcpy_d2d(nxt, cur, nbytes_valid, stream)
    lib.event_record(ev[3], stream)
    lib.stream_synchronize(stream)
    ms_copy = C.c_float()
    lib.event_elapsed_ms(ev[2], ev[3], C.byref(ms_copy))
    copy_gbs = 2 * nbytes_valid / (ms_copy.value / 10 * 1e-3) / 1e9
    cells = n**3
    # bytes ONE launch has to move: every cell read once and written once — for the two-steps-per-sweep kernel that is one
    # read + one write for TWO steps (the intermediate level lives in registers).  `frac` is priced on these moved bytes;
    # `effective_frac` prices the same launch at SURVEY.md 8d's per-cell-step figure (16 B x cell-steps per launch), i.e.
    # against a kernel that goes through HBM once per step — it may exceed 1 because temporal blocking removes bytes.
    moved_bytes = cells * BYTES_PER_CELL_STEP
    alg_bytes = cells * BYTES_PER_CELL_STEP * steps_per_launch
    achieved = moved_bytes / t_kernel / 1e9
    # PMC bytes per launch, keyed by the INSTANCE name the library reports and the grid size: a profile of another instance never labels this one
    traffic, traffic_source = None, None
    tfile = ROOT / "profiles" / "traffic.json"
    if tfile.exists():
        try:
            tj = json.loads(tfile.read_text())
            entry = (tj.get("kernels") or {}).get(f"{kernel_instance} @ {n}^3")
            if entry:
                traffic, traffic_source = entry.get("bytes_per_launch"), entry.get("source")
        except (ValueError, OSError, AttributeError):
            traffic = None
    out = {
        "wall": wall,
        "wall_stats": wall_stats,
        "ms_events": ms_events.value,
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "kernel": kernel_instance + (" - TWO fused laplace + D*, dt*, += steps per launch" if steps_per_launch == 2 else " - fused laplace + D*, dt*, +="),
            "kernel_ms": round(t_kernel * 1e3, 4), "kernel_ms_repeats": kernel_stats,
            "frac_best": round(moved_bytes / (kernel_stats["min"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "moved_bytes_per_launch": moved_bytes,
            "steps_per_launch": steps_per_launch,
            "effective_frac": round(alg_bytes / t_kernel / 1e9 / HBM_PEAK_GBS, 4),
            "effective_bytes_per_launch": alg_bytes,
            "traffic_frac": round(traffic / t_kernel / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
            "traffic_source": traffic_source,
            "copy_ceiling": round(copy_gbs, 1), "frac_of_copy_ceiling": round(achieved / copy_gbs, 4),
            "note": "frac = bytes one launch must move (1 read + 1 write per cell; two Euler steps per launch) / kernel time / peak; "
                    "effective_frac = SURVEY 8d's 16 B per cell-step x cell-steps per launch / kernel time / peak; traffic = HBM bytes per "
                    "launch from rocprofv3 PMC counters of the same build on another box (see traffic_source: a PROFILE-SOURCED constant, not "
                    "a measurement of this run), kernel time from this run (median of kernel_ms_repeats; frac_best from their minimum); "
                    "copy_ceiling = hipMemcpyDtoD of the state (the same read + write bytes) measured in this run, GB/s",
        },
        "device": backend.device_name,
    }
    t_gpu = time.perf_counter()
    out["phase_seconds_main"] = round(t_gpu - t_start, 2)
    if not args.no_extra:
        out["roofline_operator"] = operator_roofline(backend, lib, spec, cur, nxt, stream, ev, cells, max(1, args.repeats))
    del a, b
    try:
        if not args.no_extra:
            ops = more_rooflines(backend, lib, spec, stream, ev, n, min(5, max(1, args.repeats)))
            nt_gbs = ops["copy"]["nt_copy_gbs"]
            # the yardstick VERDICT r4 asked for: the kernels against the fastest copy of the same run
            out["roofline"]["nt_copy"] = nt_gbs
            out["roofline"]["frac_of_nt_copy"] = round(out["roofline"]["achieved"] / nt_gbs, 4)
            out["roofline_operator"]["nt_copy"] = nt_gbs
            out["roofline_operator"]["frac_of_nt_copy"] = round(out["roofline_operator"]["achieved"] / nt_gbs, 4)
            out["roofline_operators"] = ops
            if not args.no_configs and args.configs_scale == 1:
                out["roofline_operators"]["tile2d"] = tile2d_roofline(backend, lib, stream, ev, min(5, max(1, args.repeats)))
        out["parity"] = parity_bit(backend, n)      # (always: the digest of the state is what makes the N > 1 lines checkable)
        t_extra = time.perf_counter()
        if not args.no_extra and not args.no_configs:
            out["extra"] = extra_configs(backend, max(1, args.configs_scale))
            out["extra"]["slab_share_to_self"] = slab_share_to_self(n, wall / args.steps * 1e3, steps=400 if args.configs_scale == 1 else 8)
        out["phase_seconds"] = {"operators_and_parity_s": round(t_extra - t_gpu, 2), "extra_s": round(time.perf_counter() - t_extra, 2)}
    except Exception as err:   # the metric line must survive a failure of the side measurements; it says so
        out.setdefault("parity", None)
        out["extra_error"] = f"{type(err).__name__}: {err}"
    return out


def operator_roofline(backend, lib, spec, a, out, stream, ev, cells: int, repeats: int = 5) -> dict:
    """The kernel north_star names: the 3-D fp64 Laplacian on the resident field (`pdehip_laplace`, ghost cells set once),
    `repeats` x 60 applications timed with HIP events on the launch stream (median; every sample and the minimum reported);
    16 algorithmic bytes per cell."""
    from pde_hip import _abi

    info = spec.info
    lib.set_ghost_cells(info.ref, 1, spec.bc_c.c, a, stream)
    for _ in range(5):
        lib.laplace(info.ref, a, out, _abi.OUT_FULL, stream)
    lib.stream_synchronize(stream)
    reps, samples = 60, []
    for _ in range(repeats):
        lib.event_record(ev[2], stream)
        for _ in range(reps):
            lib.laplace(info.ref, a, out, _abi.OUT_FULL, stream)
        lib.event_record(ev[3], stream)
        lib.stream_synchronize(stream)
        ms = C.c_float()
        lib.event_elapsed_ms(ev[2], ev[3], C.byref(ms))
        samples.append(ms.value / reps)
    stats = _stats(samples)
    t = stats["median"] * 1e-3
    gbs = cells * BYTES_PER_CELL_STEP / t / 1e9
    return {"bound": "hbm", "kernel": "lap_march_kernel<double,3-D> (pdehip_laplace: one read + one write per cell)", "applications": reps,
            "kernel_ms": round(t * 1e3, 4), "kernel_ms_repeats": stats, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_best": round(cells * BYTES_PER_CELL_STEP / (stats["min"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "mcells_per_s": round(cells / t / 1e6, 1), "bytes_per_launch": cells * BYTES_PER_CELL_STEP}


def more_rooflines(backend, lib, spec, stream, ev, n: int, repeats: int = 5) -> dict:
    """HIP-event timings of the other operators of SURVEY.md 8 rows a3 / a4 / f1 at the benchmark size (VERDICT r4 "next" #8): gradient (8 B read +
    24 B written per cell), divergence (24 + 8), gradient_squared (8 + 8), the vector Laplacian (three sweeps of the scalar kernel: 3 x 16) -
    each `repeats` x 30 applications on resident arrays, median / min / every sample - and the copy yardsticks of THIS run: `pdehip_copy_nt`
    (16 bytes per thread, streaming stores: the fastest copy order of the part) next to hipMemcpyDtoD."""
    from pde_hip import _abi
    from pde_hip.device import DeviceArray

    info = spec.info
    cells = n**3
    scalar_in, scalar_out = DeviceArray(info), DeviceArray(info)
    vec_a, vec_b = DeviceArray(info, (3,)), DeviceArray(info, (3,))
    rng = np.random.default_rng(1)
    scalar_in.set_valid(rng.random((n, n, n)), stream)
    vec_a.set_valid(rng.random((3, n, n, n)), stream)
    lib.set_ghost_cells(info.ref, 1, spec.bc_c.c, scalar_in.ptr, stream)
    comp_bytes = int(info.comp_elems) * vec_a.itemsize      # bytes between the components of a full array

    def timed(fn, reps=30):
        for _ in range(3):
            fn()
        lib.stream_synchronize(stream)
        samples = []
        for _ in range(repeats):
            lib.event_record(ev[2], stream)
            for _ in range(reps):
                fn()
            lib.event_record(ev[3], stream)
            lib.stream_synchronize(stream)
            ms = C.c_float()
            lib.event_elapsed_ms(ev[2], ev[3], C.byref(ms))
            samples.append(ms.value / reps)
        return _stats(samples)

    def entry(kernel, bytes_per_cell, stats, note=None):
        t = stats["median"] * 1e-3
        gbs = cells * bytes_per_cell / t / 1e9
        out = {"bound": "hbm", "kernel": kernel, "bytes_per_cell": bytes_per_cell, "kernel_ms": round(t * 1e3, 4), "kernel_ms_repeats": stats,
               "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
        if note:
            out["note"] = note
        return out

    # the periodic ghost cells of every component are set once (the operators read them from memory)
    for k in range(3):
        lib.set_ghost_cells(info.ref, 1, spec.bc_c.c, vec_a.ptr + k * comp_bytes, stream)
    res = {}
    res["gradient"] = entry("lap_march_kernel<double, LAP_GRAD_C> (pdehip_gradient, central)", 32,
                            timed(lambda: lib.gradient(info.ref, _abi.CENTRAL, scalar_in.ptr, vec_b.ptr, _abi.OUT_FULL, stream)))
    res["divergence"] = entry("div_march_kernel<double, central> (pdehip_divergence)", 32,
                              timed(lambda: lib.divergence(info.ref, _abi.CENTRAL, vec_a.ptr, scalar_out.ptr, _abi.OUT_FULL, stream)))
    res["gradient_squared"] = entry("lap_march_kernel<double, LAP_GRADSQ_C> (pdehip_gradient_squared, central)", 16,
                                    timed(lambda: lib.gradient_squared(info.ref, 1, scalar_in.ptr, scalar_out.ptr, _abi.OUT_FULL, stream)))

    def vector_laplace():
        for k in range(3):
            lib.laplace(info.ref, vec_a.ptr + k * comp_bytes, vec_b.ptr + k * comp_bytes, _abi.OUT_FULL, stream)

    res["vector_laplace"] = entry("lap_march_kernel<double> x 3 components (vector_laplace: cartesian.py:999-1030 applies the scalar operator per component)",
                                  48, timed(vector_laplace, reps=10))
    # copy yardsticks on the same bytes as one Laplacian (1 read + 1 write per cell)
    nbytes = scalar_in.nbytes // 16 * 16
    nt = timed(lambda: lib.copy_nt(scalar_out.ptr, scalar_in.ptr, nbytes, stream))
    d2d = timed(lambda: lib.memcpy_d2d(scalar_out.ptr, scalar_in.ptr, nbytes, stream))
    res["copy"] = {"bytes": 2 * nbytes, "nt_copy_ms": nt["median"], "nt_copy_gbs": round(2 * nbytes / (nt["median"] * 1e-3) / 1e9, 1),
                   "hipMemcpyDtoD_ms": d2d["median"], "hipMemcpyDtoD_gbs": round(2 * nbytes / (d2d["median"] * 1e-3) / 1e9, 1),
                   "note": "pdehip_copy_nt: one 16-byte vector per thread in address order, streaming stores - the fastest copy order of the part "
                           "(tools/microbench4.hip); whole ghost-padded array, i.e. slightly more bytes than the cells of one sweep"}
    return res


def tile2d_roofline(backend, lib, stream, ev, repeats: int = 5) -> dict:
    """`tile2d_kernel` (2-D grids: K Euler steps per launch, time levels in LDS; BASELINE configs 2 and 3).  These grids live in the caches and
    the loop is LAUNCH-bound, not HBM-bound: reported as microseconds per launch next to the launch floor of this run (back-to-back launches of
    a kernel that moves 16 bytes), not as a fraction of the HBM peak."""
    import pde_hip
    from pde_hip.device import DeviceArray, DeviceBuffer

    # launch floor: `pdehip_copy_nt` of 16 bytes, 400 launches back to back on the same stream, HIP events
    tiny_a, tiny_b = DeviceBuffer(256), DeviceBuffer(256)
    for _ in range(50):
        lib.copy_nt(tiny_b.ptr, tiny_a.ptr, 16, stream)
    lib.stream_synchronize(stream)
    floor = []
    for _ in range(repeats):
        lib.event_record(ev[2], stream)
        for _ in range(400):
            lib.copy_nt(tiny_b.ptr, tiny_a.ptr, 16, stream)
        lib.event_record(ev[3], stream)
        lib.stream_synchronize(stream)
        ms = C.c_float()
        lib.event_elapsed_ms(ev[2], ev[3], C.byref(ms))
        floor.append(ms.value / 400 * 1e3)
    floor_us = sorted(floor)[len(floor) // 2]
    out = {"launch_floor_us": round(floor_us, 3), "launch_floor_what": "median of back-to-back launches of a 16-byte copy kernel on one stream (HIP events)"}
    for name, eq, shape, per_launch in (("cfg2_diffusion_1024sq", pde_hip.DiffusionPDE(1.0), (1024, 1024), 8), ("cfg3_cahn_hilliard_512sq", pde_hip.CahnHilliardPDE(1.0), (512, 512), 4)):
        grid = pde_hip.UnitGrid(shape, periodic=True)
        state = pde_hip.ScalarField.random_uniform(grid, -0.1, 0.1, rng=np.random.default_rng(2))
        spec = backend.make_rhs_spec(eq, state)
        a, b = DeviceArray(spec.info).set_valid(state.data, stream), DeviceArray(spec.info)
        res = C.c_void_p()
        dt, steps = (0.1, 800) if name.startswith("cfg2") else (1e-3, 800)
        lib.euler_run(spec.info.ref, spec.ref, a.ptr, b.ptr, dt, 64, C.byref(res), stream)
        lib.stream_synchronize(stream)
        samples = []
        for _ in range(repeats):
            lib.event_record(ev[2], stream)
            lib.euler_run(spec.info.ref, spec.ref, a.ptr, b.ptr, dt, steps, C.byref(res), stream)
            lib.event_record(ev[3], stream)
            lib.stream_synchronize(stream)
            ms = C.c_float()
            lib.event_elapsed_ms(ev[2], ev[3], C.byref(ms))
            samples.append(ms.value / steps)
        st = _stats(samples)
        us_launch = st["median"] * 1e3 * per_launch
        out[name] = {"bound": "launch", "kernel": "tile2d_kernel (K steps per launch, levels in LDS) through pdehip_euler_run", "steps_per_launch": per_launch,
                     "us_per_step": round(st["median"] * 1e3, 3), "us_per_launch": round(us_launch, 3), "launch_floor_us": round(floor_us, 3),
                     "launches_of_floor": round(us_launch / floor_us, 2), "ms_per_step_repeats": st,
                     "note": "the field is cache-resident: a launch advances it `steps_per_launch` steps in LDS; what bounds the loop is the launch itself"}
    return out


def slab_share_to_self(n: int, ms_per_step_n1: float, steps: int = 400, shares=(2, 4, 8)) -> dict:
    """The N-GPU slab step of THIS grid, measured on one GPU (VERDICT r5 "next" #1b): for the 1/2, 1/4 and 1/8 shares of the n^3 grid - an axis-0
    slab of n/N layers, periodic - the slab loop of `bench.py --gpus N` with its halo exchange sent TO SELF through RCCL (same C loop, same
    streams, same ncclSend / ncclRecv groups; a message is a local copy instead of an xGMI transfer), and the same slab without neighbours.
    `projected_speedup` = ms_per_step of the N = 1 line / ms_per_step of the share with exchange: what N GPUs would reach if a real link
    delivered a halo as fast as the local copy does - an upper bound from a one-GPU box, not a scaling measurement."""
    import pde_hip
    from pde_hip.distributed import SerialControl, SlabStepper

    out = {"what": "axis-0 slab shares of the benchmark grid on ONE GPU, halo exchange to self through RCCL; projected_speedup = ms_per_step(N=1) / with_exchange",
           "ms_per_step_n1": round(ms_per_step_n1, 5), "steps": steps}
    eq = pde_hip.DiffusionPDE(1.0)
    for parts in shares:
        if n % parts or n // parts < 4:
            continue
        shape = (n // parts, n, n)
        grid = pde_hip.UnitGrid(shape, periodic=True)
        entry = {"shape": list(shape)}
        for key, force in (("with_exchange", True), ("without_exchange", False)):
            st = SlabStepper(eq, grid, control=SerialControl(), force_exchange=force)
            cur, nxt = st.buf("state_a"), st.buf("state_b")
            st.set_local(cur, np.random.default_rng(0).random(shape))
            cur = st.euler_steps(cur, nxt, 0.1, 20)
            nxt = st.buf("state_b") if cur is st.buf("state_a") else st.buf("state_a")
            st.synchronize()
            best, enq = None, None
            for _ in range(3):
                t0 = time.perf_counter()
                cur = st.euler_steps(cur, nxt, 0.1, steps)
                t_enq = time.perf_counter() - t0
                st.synchronize()
                t_all = time.perf_counter() - t0
                nxt = st.buf("state_b") if cur is st.buf("state_a") else st.buf("state_a")
                if best is None or t_all < best:
                    best, enq = t_all, t_enq
            entry[key + "_ms_per_step"] = round(best / steps * 1e3, 5)
            entry[key + "_host_enqueue_us_per_step"] = round(enq / steps * 1e6, 2)
            if force:
                entry["steps_per_exchange"] = 4 if st._euler4 else (2 if st._euler2 else 1)
            st.close()
        entry["projected_speedup"] = round(ms_per_step_n1 / entry["with_exchange_ms_per_step"], 2)
        entry["projected_speedup_if_the_exchange_were_free"] = round(ms_per_step_n1 / entry["without_exchange_ms_per_step"], 2)
        out[f"1/{parts}"] = entry
    return out


def parity_bit(backend, n: int) -> dict | None:
    """Parity check accompanying the timing: 6 Euler steps from the seeded initial state through the SAME
    entry point the timed region uses (`pdehip_euler_run`: three two-step sweeps), SHA-256 of the whole final field - at 512^3
    against the digest of the REFERENCE's own torch-CPU run (tests/golden/configs.npz, written by tests/golden/make_golden_configs.py);
    at any size the digest itself is reported (`state_sha256_after_6_steps`: the N > 1 lines must show the same one)."""
    import hashlib

    import pde_hip

    grid = pde_hip.UnitGrid([n] * 3, periodic=True)
    state = pde_hip.ScalarField.random_uniform(grid, rng=np.random.default_rng(0))
    res = pde_hip.DiffusionPDE(1.0).solve(state, t_range=0.6, dt=0.1, solver="euler", backend=backend)
    digest = hashlib.sha256(np.ascontiguousarray(res.data).tobytes()).hexdigest()
    out = {"check": "sha256 of the whole field after 6 Euler steps from the seeded state", "ok": None, "sha256_full": digest}
    path = ROOT / "tests" / "golden" / "configs.npz"
    if n == 512 and path.exists():
        golden = np.load(path, allow_pickle=False)
        key = "cfg4_diffusion_512cube_6steps/sha256"
        if key in golden.files:
            out.update(check="sha256 of the whole 512^3 field after 6 Euler steps == the reference's torch-CPU run", ok=digest == str(golden[key]),
                       sha256=digest[:16], golden="tests/golden/configs.npz:cfg4_diffusion_512cube_6steps")
    return out


def extra_configs(backend, scale: int = 1) -> dict:
    """One-line timings of the other BASELINE.json configurations through `eq.solve` of the mirror front end (state uploaded
    once, resident; wall until the device is done).  They are parity-test cases (tests/test_baseline_configs.py), not the
    metric; quoted here so that the driver's run records them next to it."""
    import pde_hip

    rng = np.random.default_rng(0)
    out = {}

    def run(name, eq, grid, dtype, t_range, dt, solver, lo=0.0, hi=1.0, moved_values_per_attempt=0, **kw):
        state = pde_hip.ScalarField(grid, rng.uniform(lo, hi, grid.shape), dtype=dtype)
        # warm-up: allocations, run-time builds - and the CLOCKS: the solves of the small 2-D grids follow seconds of host work (parity digest)
        # during which the device idles; a 43 ms run of 12 us kernels from idle clocks measured 43 us per step where the same loop takes 3.4
        # straight after heavy kernels (profiles/r05_final_gpu_suite.md), so every configuration first runs at full length once, untimed
        eq.solve(state, t_range=t_range, dt=dt, solver=solver, backend=backend, **kw)
        backend.synchronize()
        # a short run (the fixed cost of a solve) and the full one, three times each, the fastest of each kind: single timings of these
        # 5-50 ms runs scatter (cfg5: 712-837 us per attempt from line to line of round 5 with one timing each)
        wall_short = wall = None
        for _ in range(3):
            t0 = time.perf_counter()
            _, short = eq.solve(state, t_range=t_range / 50, dt=dt, solver=solver, backend=backend, ret_info=True, **kw)
            backend.synchronize()
            el = time.perf_counter() - t0
            wall_short = el if wall_short is None else min(wall_short, el)
            t0 = time.perf_counter()
            _, info = eq.solve(state, t_range=t_range, dt=dt, solver=solver, backend=backend, ret_info=True, **kw)
            backend.synchronize()
            el = time.perf_counter() - t0
            wall = el if wall is None else min(wall, el)
        steps = info["solver"]["steps"]
        cells = int(np.prod(grid.shape))
        out[name] = {"steps": steps, "wall_ms": round(wall * 1e3, 2), "us_per_step": round(wall / steps * 1e6, 2),
                     "mcell_steps_per_s": round(cells * steps / wall / 1e6, 1)}
        # the same without the fixed cost of a solve (stepper construction, upload, download): long run minus short run
        s_short = int(short["solver"]["steps"])
        if steps > s_short and wall > wall_short:
            out[name]["us_per_step_net"] = round((wall - wall_short) / (steps - s_short) * 1e6, 2)
        attempts = info["solver"].get("attempts")
        if attempts and moved_values_per_attempt:
            # an RKF45 attempt of the fused stage sweeps moves, per cell: stages 1-5 read the stage input, y and the earlier slopes and write
            # the slope and the next input (4 + 5 + 6 + 7 + 8 values), the last stage reads input, y, k1, k3, k4, k5 and writes the new state
            # (7): 37 values - SURVEY 8d's schedule with separate combination kernels counts 56 (224 B in fp32)
            # differential timing: (long run - short run) / (attempts of the long run - attempts of the short one) takes the fixed cost of a
            # solve (stepper construction, upload, download) out of the attempt
            a_short = int(short["solver"].get("attempts") or 0)
            t_attempt = (wall - wall_short) / max(1, attempts - a_short) if attempts > a_short else wall / attempts
            moved = cells * moved_values_per_attempt * np.dtype(dtype).itemsize
            out[name].update(attempts=int(attempts), us_per_attempt=round(t_attempt * 1e6, 2), moved_bytes_per_attempt=int(moved),
                             frac_of_peak_on_moved_bytes=round(moved / t_attempt / 1e9 / HBM_PEAK_GBS, 4),
                             frac_of_peak_on_survey_bytes=round(cells * 56 * np.dtype(dtype).itemsize / t_attempt / 1e9 / HBM_PEAK_GBS, 4))

    # (`scale` > 1: the same solves on smaller grids and shorter runs - a test aid for the host shim, tests/test_distributed_gloo.py)
    run("cfg2_diffusion_1024sq_f64_euler", pde_hip.DiffusionPDE(), pde_hip.CartesianGrid([[0, 1024 // scale]] * 2, 1024 // scale, periodic=True), np.float64,
        100.0 / scale, 0.1, "euler")
    run("cfg3_cahn_hilliard_512sq_f64_euler_1e4_steps", pde_hip.CahnHilliardPDE(), pde_hip.UnitGrid([512 // scale, 512 // scale]), np.float64, 10.0 / scale, 1e-3, "euler")
    # cfg5's expression is recognised as the Cahn-Hilliard FORM (pde_hip/backend.py `_match_expression_rhs`) and runs the fused two-level
    # sweeps of that class, NOT the generic expression compiler; the line next to it is a two-pass expression of the same cost that is
    # not of that form (one more term) and goes through the run-time compiled passes (pdehip_jit_rk_run)
    run("cfg5_expression_256cube_f32_rkf45_fused_CH_form", pde_hip.PDE({"c": "laplace(c**3 - c - laplace(c))"}), pde_hip.UnitGrid([256 // scale] * 3, periodic=True),
        np.float32, 1.0, 1e-3, "runge-kutta", lo=-0.1, hi=0.1, moved_values_per_attempt=37, adaptive=True)
    run("generic_two_pass_expression_256cube_f32_rkf45", pde_hip.PDE({"c": "laplace(c**3 - c - laplace(c)) - 0.01 * c"}), pde_hip.UnitGrid([256 // scale] * 3, periodic=True),
        np.float32, 1.0, 1e-3, "runge-kutta", lo=-0.1, hi=0.1, adaptive=True)
    # the headline grid with walls whose conditions depend on time and position on all six faces (tools/time_bc_program.py): two
    # steps per sweep with a second coefficient set + the cells next to the faces recomputed (csrc/pdehip_shell.hip); differential timing of two
    # run lengths (upload, download and run-time builds cancel)
    n = 512 // scale
    grid = pde_hip.CartesianGrid([[0, 1]] * 3, [n] * 3, periodic=False)
    dt = 0.1 * float(grid.discretization[0]) ** 2
    bc = {"x-": {"value_expression": "0.2 * sin(3 * t) + 0.05 * y"}, "x+": {"derivative_expression": "0.1 * cos(t) * z"},
          "y-": {"value_expression": "x * z * (1 + t)"}, "y+": {"derivative_expression": "0.05 * x * sin(t)"},
          "z-": {"value_expression": "tanh(x - y) * t"}, "z+": {"derivative_expression": "0.1 * y * cos(2 * t)"}}
    eq = pde_hip.DiffusionPDE(1.0, bc=bc)
    state = pde_hip.ScalarField(grid, rng.uniform(-1, 1, grid.shape))
    eq.solve(state, 4 * dt, dt, solver="euler", backend=backend)

    def timed(count):
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            res = eq.solve(state, count * dt, dt, solver="euler", backend=backend)
            float(res.data[0, 0, 0])
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        return best

    t1, t2 = (timed(100), timed(300)) if scale == 1 else (timed(10), timed(210))
    out["diffusion_512cube_f64_euler_walls_of_time_and_position"] = {"us_per_step": round((t2 - t1) / 200 * 1e6, 2),
                                                                     "mcell_steps_per_s": round(n**3 * 200 / (t2 - t1) / 1e6, 1)}
    return out


def bench_distributed(args) -> dict:
    """One rank per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Data plane: libpdehip +
    RCCL over xGMI only; torch.distributed (gloo) is the control plane: RCCL id, agreeing on code paths, barriers.

    The field is the SAME seeded global field the N = 1 line starts from (`ScalarField.random_uniform(grid, rng=default_rng(0))`), cut
    per rank.  Besides the timed region (`--repeats` times EXACTLY K steps between barrier + synchronize, MAX over ranks per repetition)
    the line carries: `parity` - SHA-256 of the gathered field after 6 steps from that state against the digest of the REFERENCE's own
    run (512^3) -, `state_sha256` at any size (equal to the N = 1 line's: tests/test_distributed_gloo.py), per rank the time of the same
    steps on its slab WITHOUT neighbours (`compute_only_ms_per_step`: the serial loop on a grid of the slab's shape) and the difference
    `exchange_exposed_ms_per_step`, and the `roofline` of the slab sweep priced on the compute-only time."""
    import hashlib

    import pde_hip
    from pde_hip.distributed import SerialControl, SlabStepper, TorchControl

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    control = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=world)
        control = TorchControl()
    n = args.size
    grid = pde_hip.UnitGrid([n, n, n], periodic=True)
    eq = pde_hip.DiffusionPDE(1.0)
    blocks = args.decomposition != "slab"
    if blocks:
        from pde_hip.distributed import BlockStepper
        from pde_hip.mesh import block_decomposition

        dims = block_decomposition(grid.shape, world) if args.decomposition == "auto" else [int(v) for v in args.decomposition.split(",")]
        stepper = BlockStepper(eq, grid, dims=dims, control=control, device=local_rank, force_exchange=args.force_distributed)
    else:
        stepper = SlabStepper(eq, grid, control=control, device=local_rank, force_exchange=args.force_distributed)
    control = stepper.control
    # synthetic data: the seeded global field of the N = 1 line; every rank keeps its slab only
    local0 = np.ascontiguousarray(stepper.mesh.extract(pde_hip.ScalarField.random_uniform(grid, rng=np.random.default_rng(0)).data))
    a, b = stepper.buf("state_a"), stepper.buf("state_b")
    dt = 0.1

    def gather_to_rank0(buf):
        # raw buffers, every part crosses the control plane once (pde_hip.distributed.gather_parts)
        return stepper.gather(buf, root=0)

    # parity: 6 steps (three two-step sweeps with their exchanges) from the seeded state, whole field hashed on rank 0
    stepper.set_local(a, local0)
    res = stepper.euler_steps(a, b, dt, 6)
    stepper.synchronize()
    final6 = gather_to_rank0(res)
    parity = None
    digest = hashlib.sha256(np.ascontiguousarray(final6).tobytes()).hexdigest() if rank == 0 else None
    del final6
    gpath = ROOT / "tests" / "golden" / "configs.npz"
    if rank == 0 and n == 512 and gpath.exists():
        golden = np.load(gpath, allow_pickle=False)
        key = "cfg4_diffusion_512cube_6steps/sha256"
        if key in golden.files:
            parity = {"check": "sha256 of the gathered 512^3 field after 6 slab-parallel Euler steps == the reference's torch-CPU run",
                      "ok": digest == str(golden[key]), "sha256": digest[:16], "golden": "tests/golden/configs.npz:cfg4_diffusion_512cube_6steps"}
    # warm-up, then the timed region: EXACTLY K steps between barrier + synchronize, `--repeats` times
    stepper.set_local(a, local0)
    cur = stepper.euler_steps(a, b, dt, args.warmup)
    nxt = b if cur is a else a
    stepper.synchronize()
    walls_max, walls_own = [], []
    for _ in range(max(1, args.repeats)):
        stepper.synchronize()
        control.barrier()
        t0 = time.perf_counter()
        out = stepper.euler_steps(cur, nxt, dt, args.steps)
        stepper.synchronize()
        own = time.perf_counter() - t0
        control.barrier()
        walls_own.append(own)
        walls_max.append(max(control.allgather(time.perf_counter() - t0)))
        if out is not cur:
            cur, nxt = nxt, cur
    ok = all(control.allgather(bool(np.isfinite(stepper.gather_local(cur)).all())))
    if blocks:
        info = {"decomposition": [int(d) for d in stepper.dims], "two_steps_per_sweep": bool(stepper.block2), "fast_block_loop": bool(stepper.block2),
                "cells_per_rank": [int(v) for v in stepper.mesh.local_shape]}
    else:
        info = {"two_steps_per_sweep": stepper._euler2, "steps_per_exchange": 4 if stepper._euler4 else (2 if stepper._euler2 else 1),
                "layers_per_rank": [int(c) for c in stepper.mesh.counts]}
    # the same steps on this rank's slab WITHOUT neighbours: the serial loop (two steps per sweep) on a periodic grid of the slab's shape
    local_shape = [int(v) for v in stepper.mesh.local_shape]
    serial = SlabStepper(eq, pde_hip.UnitGrid(local_shape, periodic=True), control=SerialControl(), device=local_rank)
    sa, sb = serial.buf("state_a"), serial.buf("state_b")
    serial.set_local(sa, local0)
    sc = serial.euler_steps(sa, sb, dt, max(2, args.warmup))
    serial.synchronize()
    alone = []
    for _ in range(max(1, args.repeats)):
        other = sb if sc is sa else sa
        t0 = time.perf_counter()
        sc = serial.euler_steps(sc, other, dt, args.steps)
        serial.synchronize()
        alone.append((time.perf_counter() - t0) / args.steps * 1e3)
    serial.close()
    own_stats, alone_stats = _stats([w / args.steps * 1e3 for w in walls_own]), _stats(alone)
    per_rank = control.allgather({"rank": rank, "layers": local_shape[0], "ms_per_step": own_stats["median"],
                                  "compute_only_ms_per_step": alone_stats["median"],
                                  "exchange_exposed_ms_per_step": round(own_stats["median"] - alone_stats["median"], 5)})
    # what RCCL itself reports (VERDICT r5 "next" #8b): ranks in the communicator, its version, the device and PCI bus id behind every rank
    rccl = None
    if getattr(stepper, "comm", None) is not None:
        five, bus = (C.c_int * 5)(), C.create_string_buffer(64)
        stepper.lib.comm_info(stepper.comm, five, bus, 64)
        mine = {"rank": rank, "nccl_comm_count": int(five[0]), "nccl_user_rank": int(five[1]), "nccl_device": int(five[2]), "hip_device": int(five[4]),
                "pci_bus_id": bus.value.decode(errors="replace")}
        ranks = control.allgather(mine)
        rccl = {"nranks": int(five[0]), "version": int(five[3]), "ranks": ranks,
                "distinct_devices": len({(r["pci_bus_id"], r["hip_device"]) for r in ranks})}
    stepper.close()
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()
    wall_stats = _stats([w / args.steps * 1e3 for w in walls_max])
    local_cells = int(np.prod(local_shape))
    t_alone = alone_stats["median"] * 1e-3
    roofline = {"bound": "hbm", "kernel": "slab sweep of this rank without neighbours (euler2_kernel: two steps per launch, 1 read + 1 write per cell)",
                "moved_bytes_per_step_pair": local_cells * BYTES_PER_CELL_STEP, "ms_per_step": alone_stats["median"],
                "achieved": round(local_cells * BYTES_PER_CELL_STEP / (2 * t_alone) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(local_cells * BYTES_PER_CELL_STEP / (2 * t_alone) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                "note": "per GPU, rank 0; wall-clock of K steps of the serial two-step loop on a grid of the slab's shape (no HIP events: the "
                        "loop runs on the library's own streams); the exchange and its choreography are what `per_rank` shows on top"}
    return {"wall": wall_stats["median"] * 1e-3 * args.steps, "wall_stats": wall_stats, "rank": rank, "finite": ok, "info": info, "parity": parity,
            "state_sha256": digest, "per_rank": per_rank, "roofline": roofline, "rccl": rccl}


def _free_port() -> int:
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def spawn_ranks(n: int) -> int:
    """`bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): this process becomes the launcher - one child
    per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what `torch.distributed.run` sets), the children's output passed through
    (rank 0 prints the JSON line), non-zero exit if any rank fails.  The reference's MPI entry is launcher-driven as well
    (pde/solvers/explicit_mpi.py:133-226: `mpiexec -n N`); the bench contract is `--gpus N`, so a bare invocation fans out itself."""
    import subprocess

    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "PDEHIP_BENCH_SPAWNED": "1"})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    procs = []
    for rank in range(n):
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]],
                                      env={**env, "RANK": str(rank), "LOCAL_RANK": str(rank), "LOCAL_WORLD_SIZE": str(n)}))
    rc = 0
    try:
        while procs:
            for pr in list(procs):
                code = pr.poll()
                if code is None:
                    continue
                procs.remove(pr)
                if code != 0 and rc == 0:
                    rc = code
                    for other in procs:      # one rank failed: the others would wait in a collective for ever
                        other.terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            pr.kill()
    return rc


def main():
    args = parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # never print a line whose n_gpus differs from what was asked for
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    n = args.size
    cells = n**3
    if args.gpus > 1 or world > 1 or args.force_distributed:
        r = bench_distributed(args)
        if r["rank"] != 0:
            return
        ngpu = world
        wall = r["wall"]
        line = {"roofline": r["roofline"], "cpu_baseline": None, "slab": r["info"], "finite": r["finite"], "parity": r["parity"],
                "state_sha256_after_6_steps": r["state_sha256"], "per_rank": r["per_rank"], "rccl": r["rccl"]}
        if "decomposition" in r["info"]:
            parallelism = (f"blocks{'x'.join(str(d) for d in r['info']['decomposition'])} (boxes with two-layer halos incl. edges, one RCCL message per "
                           "neighbouring rank, rim kernel + interior sweep)")
        else:
            parallelism = f"slab{ngpu} (axis-0 slabs, RCCL send/recv halo exchange overlapped with interior kernel)"
    else:
        r = bench_single(args)
        ngpu = 1
        wall = r["wall"]
        line = {"roofline": r["roofline"]}
        for key in ("roofline_operator", "roofline_operators", "parity", "extra", "extra_error"):
            if key in r:
                line[key] = r[key]
        if isinstance(line.get("parity"), dict):
            line["state_sha256_after_6_steps"] = line["parity"].pop("sha256_full")
            if line["parity"]["ok"] is None:
                line["parity"] = None        # no reference digest at this size
        t_cpu = time.perf_counter()
        line["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(n, args.cpu_seconds)
        # wall seconds per phase of this process (VERDICT r4 "next" #8: makes the driver's clock around the run auditable)
        line["phase_seconds"] = {"import_upload_timed_region_dominant_kernel_s": r.get("phase_seconds_main"), **(r.get("phase_seconds") or {}),
                                 "cpu_baseline_s": round(time.perf_counter() - t_cpu, 2)}
        ref_file = ROOT / "profiles" / "reference_cpu.json"
        if ref_file.exists() and not args.no_cpu_baseline:
            # the reference itself (py-pde, eager torch-CPU backend) cannot travel to the GPU box: timed in the build container
            # by tools/time_reference_cpu.py next to the oracle on the same cores, committed under profiles/
            try:
                line["cpu_baseline_reference"] = json.loads(ref_file.read_text())
            except (ValueError, OSError):
                pass
        parallelism = "single GPU"
    value = cells * args.steps / wall / 1e6
    stats = r["wall_stats"]
    line["repeats"] = {**stats, "unit": "ms_per_step", "what": "every repetition is one timed region of EXACTLY `steps` steps between synchronisations; "
                                                             "value / ms_per_step are the median, value_best the minimum"}
    line["value_best"] = round(cells / (stats["min"] * 1e-3) / 1e6, 1)
    out = {
        "metric": "Mcells/s (and % HBM roofline) for 3D DiffusionPDE 512^3 fp64; 1/2/4/8-GPU scaling",
        "value": round(value, 1),
        "unit": "Mcells/s",
        "n_gpus": ngpu,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"DiffusionPDE(D=1) on UnitGrid([{n}]*3, periodic=True) fp64, explicit Euler dt=0.1, "
                        "state resident in HBM, BCs on the fly + fused laplace/update, two steps per kernel sweep (bit-identical to single steps)",
            "cells": cells,
            "parallelism": parallelism,
            "effective_hbm_roofline_frac_whole_step": round(value * 1e6 * BYTES_PER_CELL_STEP / 1e9 / (HBM_PEAK_GBS * ngpu), 4),
        },
        **line,
    }
    # RCCL writes its version banner through C stdio; drain that buffer first so the JSON line is last
    try:
        C.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
