#!/bin/bash
# round 6, call 10: halo stream priority against hardware-queue collisions (the 1/8 share after 0..3 dummy streams); fastmath tests
mkdir -p gpurun_out
export TMPDIR=/tmp
cat > /tmp/q.py <<'PY'
import sys, ctypes as C
sys.path[:0] = [".", "py-pde_amd"]
import bench, pde_hip
from pde_hip._lib import require_device
lib = require_device()
keep = []
for _ in range(int(sys.argv[1])):
    s = C.c_void_p(); lib.stream_create(C.byref(s)); keep.append(s)
r = bench.slab_share_to_self(512, 0.2, steps=400, shares=(8, 4))
print("dummy streams", sys.argv[1], {k: (v["with_exchange_ms_per_step"], v["without_exchange_ms_per_step"]) for k, v in r.items() if isinstance(v, dict)}, flush=True)
PY
{
for pr in 0 -1 1; do for d in 0 1 2 3; do PDEHIP_HALO_PRIORITY=$pr python /tmp/q.py $d 2>&1 | grep "dummy streams" | sed "s/^/halo priority $pr: /"; done; done
} | tee gpurun_out/r06_call10_priority.log
echo "== fastmath tests"; timeout 1200 python -m pytest tests/test_hip_fastmath.py -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
