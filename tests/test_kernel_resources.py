"""Guard: no kernel of the library spills to scratch memory (CPU-only check of the built library's code-object metadata).

Dynamic indexing of the kernel-argument struct or an oversized register tile silently turns into
scratch memory and halves the bandwidth of the hot kernel (happened once in round 1: 552 B/lane after
the body was moved into a device function).  The per-kernel scratch size is part of the code objects' metadata.
"""

from __future__ import annotations

import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "py-pde_amd" / "csrc"


LIB = ROOT / "py-pde_amd" / "lib" / "libpdehip.so"
LLVM_BIN = Path("/opt/rocm/lib/llvm/bin")


def _kernel_metadata(tmp_path):
    """(name, scratch bytes per lane, VGPRs) of every kernel in the BUILT library: the code objects are unbundled from the
    .so and their AMDGPU metadata notes read - seconds, where recompiling the stencil translation unit takes minutes."""
    work = tmp_path / "lib.so"
    shutil.copy(LIB, work)
    subprocess.run([str(LLVM_BIN / "llvm-objdump"), "--offloading", str(work)], capture_output=True, text=True, check=True, timeout=300)
    out = []
    for co in sorted(tmp_path.glob("lib.so.*gfx950*")):
        notes = subprocess.run([str(LLVM_BIN / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True, check=True, timeout=300).stdout
        for block in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block)
            scratch = re.search(r"\.private_segment_fixed_size:\s+(\d+)", block)
            vgpr = re.search(r"\.vgpr_count:\s+(\d+)", block)
            if name and scratch and vgpr:
                out.append((name.group(1), int(scratch.group(1)), int(vgpr.group(1))))
    return out


def test_no_scratch_spills(tmp_path):
    if not LIB.exists() or not (LLVM_BIN / "llvm-objdump").exists():
        pytest.skip("built library or llvm tools not available")
    kernels = _kernel_metadata(tmp_path)
    names = {n for n, _, _ in kernels}
    # the translation units with the register-heavy kernels are all there
    for needle in ("lap_march_kernel", "euler2_kernel", "tile2d_kernel", "div_march_kernel", "lincomb_kernel", "ghost_kernel"):
        assert any(needle in n for n in names), f"no {needle} in the library's code objects"
    assert len(kernels) > 400
    offenders = [(n, s) for n, s, _ in kernels if s]
    assert not offenders, f"kernels spilling to scratch: {offenders[:5]}"
    # the hot instances keep two waves per SIMD
    hot = [v for n, _, v in kernels if "euler2_kernel" in n]
    assert hot and max(hot) <= 256
