"""N > 1 ranks of the product's slab-parallel path on REAL GPUs (skipped where fewer devices are visible).

Launches ``tests/multirank_worker.py`` exactly like the driver launches ``bench.py`` (``python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1``): diffusion Euler (one-step loop and two steps per sweep), the
fused Cahn-Hilliard sweep, RK4 and adaptive RKF45 with the MAX all-reduce — gathered result == serial oracle bit for bit,
equal step counts.  The subprocess runs under a watchdog timeout, so a mismatched send/recv fails instead of hanging the
suite.  The same worker runs on CPU against the host shim in tests/test_distributed_gloo.py.
"""

from __future__ import annotations

import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _device_count() -> int:
    from pde_hip import _lib

    return _lib.device_count()


def launch_worker(world: int, env_extra=None, timeout: float = 600, args=()):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "multirank_worker.py"), *args]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=str(ROOT))
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("MULTIRANK ")]
    assert lines, f"worker produced no report (rc {proc.returncode}):\n{proc.stdout[-2000:]}\n{proc.stderr[-4000:]}"
    report = json.loads(lines[-1][len("MULTIRANK "):])
    assert proc.returncode == 0 and not report["failures"], f"{report['failures']}\n{proc.stderr[-2000:]}"
    return report


@pytest.mark.parametrize("world", [2, 4, 8])
def test_slab_solves_equal_serial_oracle(world):
    if _device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    report = launch_worker(world)
    assert report["world"] == world and len(report["cases"]) >= 6
    assert report["cases"]["diffusion_euler_periodic"]["two_steps_per_sweep"]
    assert report["cases"]["cahn_hilliard_euler"]["flags"] & 1


@pytest.mark.parametrize("world", [2, 8])
def test_bench_runs_on_n_gpus(world):
    """`bench.py --gpus N` as the driver launches it: prints ONE JSON line with n_gpus = N."""
    if _device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "20", "--warmup", "4", "--size", "256"]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert proc.returncode == 0 and len(line) == 1, proc.stderr[-3000:]
    out = json.loads(line[0])
    assert out["n_gpus"] == world and out["finite"] and out["value"] > 0
