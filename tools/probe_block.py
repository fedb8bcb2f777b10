"""Overhead probe for the block steppers on ONE GPU: a block-sized periodic grid (256 x 128 x 512 = the share of one GPU of 512^3 on
2 x 4 x 1 blocks; 256^3 = the share on 2 x 2 x 2) whose halos are sent to the block itself through pack -> RCCL -> unpack (the code path
of N ranks), against the same block without any exchange (the serial two-steps-per-sweep loop).
  fast : csrc/pdehip_block2_loops.h (two steps per sweep, one message per peer, exchange hidden behind the next sweep, rim recomputed)
  old  : csrc/pdehip_block_loops.h (PDEHIP_BLOCK2=0: one step per sweep, exchange in front of every sweep)
usage: python tools/probe_block.py 256,128,512 [steps]"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import numpy as np

import pde_hip
from pde_hip.distributed import BlockStepper

shape = tuple(int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "256,128,512").split(","))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
grid = pde_hip.UnitGrid(shape, periodic=True)
eq = pde_hip.DiffusionPDE(1.0)
cells = int(np.prod(shape))
rows = []
configs = (("fast loop, halos to self", True, "1"), ("no exchange", False, "1"), ("old loop, halos to self", True, "0"))
if os.environ.get("PROBE_ONLY"):      # (a kernel trace of one configuration: PROBE_ONLY=0 / 1 / 2)
    configs = (configs[int(os.environ["PROBE_ONLY"])],)
for label, force, fast in configs:
    os.environ["PDEHIP_BLOCK2"] = fast
    st = BlockStepper(eq, grid, force_exchange=force)
    if force and st.block2 and os.environ.get("PROBE_CUT"):      # e.g. PROBE_CUT=1,0,0: only the first axis travels (a slab through the block loop)
        import ctypes as C

        st.cut[:] = [int(v) for v in os.environ["PROBE_CUT"].split(",")]
        st._cut3 = (C.c_int * 3)(*st.cut)
    cur, nxt = st.scatter(np.random.default_rng(0).random(shape)), st.buf("state_b")
    cur = st.euler_steps(cur, nxt, 0.1, 20)
    nxt = st.buf("state_b") if cur is st.buf("state_a") else st.buf("state_a")
    st.synchronize()
    best, enq = 1e9, 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        cur = st.euler_steps(cur, nxt, 0.1, steps)
        nxt = st.buf("state_b") if cur is st.buf("state_a") else st.buf("state_a")
        t_enq = time.perf_counter() - t0
        st.synchronize()
        t_all = time.perf_counter() - t0
        if t_all < best:
            best, enq = t_all, t_enq
    print(f"{shape} {label:26s} (block2={st.block2}, cut={list(st.cut)}): {best/steps*1e3:.4f} ms/step ({cells*steps/best/1e9:.1f} Gcells/s), host enqueue {enq/steps*1e6:.1f} us/step", flush=True)
    rows.append(best / steps)
    st.close()
if len(rows) == 3:
    print(f"{shape} with exchange / without: fast {rows[0]/rows[1]:.2f} x, old {rows[2]/rows[1]:.2f} x", flush=True)
