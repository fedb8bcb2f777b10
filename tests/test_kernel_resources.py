"""Guard: no kernel of the library spills to scratch memory (CPU-only check through hipcc remarks).

Dynamic indexing of the kernel-argument struct or an oversized register tile silently turns into
`ScratchSize > 0` and halves the bandwidth of the hot kernel (happened once in round 1: 552 B/lane after
the body was moved into a device function).  hipcc reports the per-kernel scratch size at compile time.
"""

from __future__ import annotations

import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "py-pde_amd" / "csrc"


@pytest.mark.parametrize("source", ["pdehip_kernels.hip", "pdehip_ops.hip"])
def test_no_scratch_spills(source, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("hipcc not available")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", f"-I{ROOT / 'py-pde_amd' / 'build'}",
           "-c", str(CSRC / source), "-o", str(tmp_path / "k.o"), "-Rpass-analysis=kernel-resource-usage"]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600, check=True)
    names = re.findall(r"Function Name: (\S+)", proc.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", proc.stderr)]
    assert names and len(names) == len(scratch)
    offenders = [(n, s) for n, s in zip(names, scratch) if s]
    assert not offenders, f"kernels spilling to scratch: {offenders[:5]}"
