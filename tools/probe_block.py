"""Overhead probe for the block stepper on ONE GPU: a block-sized periodic grid (256^3 = the 8-GPU share of 512^3) whose six faces
are sent to self through pack -> RCCL -> unpack (the code path of N ranks), against the same block without any exchange.
usage: python tools/probe_block.py 256,256,256 [steps]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "py-pde_amd")]
import numpy as np

import pde_hip
from pde_hip.distributed import BlockStepper

shape = tuple(int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "256,256,256").split(","))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
grid = pde_hip.UnitGrid(shape, periodic=True)
eq = pde_hip.DiffusionPDE(1.0)
cells = int(np.prod(shape))
for force in (True, False):
    st = BlockStepper(eq, grid, force_exchange=force)
    cur, nxt = st.scatter(np.random.default_rng(0).random(shape)), st.buf("state_b")
    cur = st.euler_steps(cur, nxt, 0.1, 20)
    nxt = st.buf("state_b") if cur is st.buf("state_a") else st.buf("state_a")
    st.synchronize()
    t0 = time.perf_counter()
    cur = st.euler_steps(cur, nxt, 0.1, steps)
    t_enq = time.perf_counter() - t0
    st.synchronize()
    t_all = time.perf_counter() - t0
    st.close()
    print(f"{shape} block stepper exchange={force}: {t_all/steps*1e3:.4f} ms/step ({cells*steps/t_all/1e9:.1f} Gcells/s), host enqueue {t_enq/steps*1e6:.1f} us/step", flush=True)
