#!/bin/bash
# round 6, call 9: (1) is the slow 1/8 slab share inside bench.py a collision of streams on hardware queues?  the share after creating 1..7 dummy
# streams, and with GPU_MAX_HW_QUEUES=8; (2) the fastmath tests + A/B timing of the contracted build
mkdir -p gpurun_out
export TMPDIR=/tmp
cat > /tmp/q.py <<'PY'
import sys, ctypes as C
sys.path[:0] = [".", "py-pde_amd"]
import bench, pde_hip
from pde_hip._lib import require_device
lib = require_device()
n_dummy = int(sys.argv[1])
keep = []
for _ in range(n_dummy):
    s = C.c_void_p(); lib.stream_create(C.byref(s)); keep.append(s)
    lib.copy_nt  # noqa
r = bench.slab_share_to_self(512, 0.2, steps=400, shares=(8,))
print("dummy streams", n_dummy, {k: (v["with_exchange_ms_per_step"], v["without_exchange_ms_per_step"]) for k, v in r.items() if isinstance(v, dict)}, flush=True)
PY
{
for d in 0 1 2 3 4 5 6 7; do python /tmp/q.py $d 2>&1 | grep "dummy streams"; done
for d in 0 2 3 5; do GPU_MAX_HW_QUEUES=8 python /tmp/q.py $d 2>&1 | grep "dummy streams" | sed 's/^/GPU_MAX_HW_QUEUES=8 /'; done
} | tee gpurun_out/r06_call09_queues.log
echo "== fastmath tests"; timeout 1200 python -m pytest tests/test_hip_fastmath.py tests/test_hip_operators.py tests/test_hip_euler2.py -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
{
for rep in 1 2; do
  for n in 512 256; do
    echo "-- $n exact"; timeout 200 python tools/time_euler2.py $n 200 2>&1 | tail -1
    echo "-- $n fastmath"; PDEHIP_FASTMATH=1 timeout 200 python tools/time_euler2.py $n 200 2>&1 | tail -1
    echo "-- $n walls exact"; TIME_PERIODIC=0 timeout 200 python tools/time_euler2.py $n 200 2>&1 | tail -1
    echo "-- $n walls fastmath"; TIME_PERIODIC=0 PDEHIP_FASTMATH=1 timeout 200 python tools/time_euler2.py $n 200 2>&1 | tail -1
  done
done
} | tee gpurun_out/r06_call09_fastmath_ab.log
