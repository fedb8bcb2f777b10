"""Stress of the world-size-1 slab stepper on tiny grids (hunting a rare all-zero result seen once in tests/pypde_slab_worker.py,
fuzz15: 1-D periodic Cahn-Hilliard, 8 cells, Euler): many fresh steppers, upload -> 4 steps -> download twice, every stage checked."""

from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "py-pde_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))
import pde_hip  # noqa: E402
from pde_hip.distributed import SlabStepper  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
if len(sys.argv) > 2 and sys.argv[2] == "gloo":      # as in the worker: torch loaded, control plane over gloo
    import os

    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29633")
    dist.init_process_group("gloo", rank=0, world_size=1)
CASES = {
    "ch1d_8": (lambda: pde_hip.CahnHilliardPDE(0.917), lambda: pde_hip.UnitGrid([8], periodic=True), 1e-3),
    "ch1d_13": (lambda: pde_hip.CahnHilliardPDE(1.239), lambda: pde_hip.CartesianGrid([[0, 26]], [13], periodic=True), 1.6e-2),
    "diff1d_8": (lambda: pde_hip.DiffusionPDE(0.7), lambda: pde_hip.UnitGrid([8], periodic=True), 1e-3),
    "ch2d": (lambda: pde_hip.CahnHilliardPDE(0.9), lambda: pde_hip.UnitGrid([8, 6], periodic=True), 1e-3),
    "diff3d": (lambda: pde_hip.DiffusionPDE(0.7), lambda: pde_hip.UnitGrid([12, 4, 6], periodic=[False, True, False]), 1e-3),
}
expect: dict[str, np.ndarray] = {}
report = {name: {"runs": 0, "upload_bad": 0, "result_bad": 0, "zero_result": 0, "nan": 0} for name in CASES}
details = []
rng = np.random.default_rng(3)
for it in range(N):
    for name, (mk_eq, mk_grid, dt) in CASES.items():
        eq, grid = mk_eq(), mk_grid()
        data = np.random.default_rng(3).uniform(-0.4, 0.4, grid.shape)
        st = SlabStepper(eq, grid)
        a, b = st.buf("state_a"), st.buf("state_b")
        cur = data
        for half in range(2):
            st.set_local(a, cur)
            up = st.gather_local(a)
            if not np.array_equal(up, cur):
                report[name]["upload_bad"] += 1
                details.append((it, name, half, "upload", float(np.abs(up).max())))
            res = st.euler_steps(a, b, dt, 4, 4 * half * dt)
            cur = st.gather_local(res)
        st.close()
        r = report[name]
        r["runs"] += 1
        if name not in expect:
            expect[name] = cur.copy()
        if not np.array_equal(cur, expect[name]):
            r["result_bad"] += 1
            r["zero_result"] += int(not cur.any())
            r["nan"] += int(not np.isfinite(cur).all())
            if len(details) < 20:
                details.append((it, name, "result", float(np.abs(cur).max()), float(np.abs(cur - expect[name]).max())))
# the first results against the single-GPU loops of the mirror API
for name, (mk_eq, mk_grid, dt) in CASES.items():
    eq, grid = mk_eq(), mk_grid()
    state = pde_hip.ScalarField(grid, np.random.default_rng(3).uniform(-0.4, 0.4, grid.shape))
    mid = eq.solve(state, 4 * dt, dt, solver="euler")
    ref = eq.solve(pde_hip.ScalarField(grid, np.array(mid.data)), 4 * dt, dt, solver="euler")
    report[name]["first_equals_single_gpu"] = bool(np.array_equal(ref.data, expect[name]))
print("STRESS1D " + json.dumps({"iterations": N, "report": report, "details": details[:20]}))
