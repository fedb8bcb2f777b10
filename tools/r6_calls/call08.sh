#!/bin/bash
# round 6, call 8: (1) the GPU suite with its summary kept; (2) why is the 1/8 slab share slower inside bench.py (0.054 ms per step) than in
# tools/probe_slab.py (0.041)?  the same function alone, in both orders; (3) sizes table on the current build
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r06_call08_gpu_suite.log 2>&1; grep -E "passed|failed|error" gpurun_out/r06_call08_gpu_suite.log | tail -3
python - <<'PY' 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06_call08_slab_share.log
import sys, json
sys.path[:0] = [".", "py-pde_amd"]
import bench
for shares in ((8,), (2, 4, 8), (8, 4, 2), (8,)):
    r = bench.slab_share_to_self(512, 0.2, steps=400, shares=shares)
    print(shares, {k: (v["with_exchange_ms_per_step"], v["without_exchange_ms_per_step"]) for k, v in r.items() if isinstance(v, dict)}, flush=True)
PY
timeout 900 python tools/time_sizes.py 512x512x512 513x513x513 515x515x515 500x500x300 512x512x513 511x511x511 510x510x510 2>/dev/null | tee gpurun_out/r06_call08_time_sizes.md
