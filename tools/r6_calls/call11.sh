#!/bin/bash
# round 6, call 11: the queue probe (compute / halo stream on different hardware queues, measured at run time) - the 1/8 and 1/4 shares after
# 0..5 dummy streams; distributed GPU tests; the bench line
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
for d in 0 1 2 3 4 5; do python /tmp/q.py $d 2>&1 | grep "dummy streams" | sed "s/^/probe on: /"; done
for d in 1 3; do PDEHIP_QUEUE_PROBE=0 python /tmp/q.py $d 2>&1 | grep "dummy streams" | sed "s/^/probe off: /"; done
} | tee gpurun_out/r06_call11_queue_probe.log
echo "== distributed tests"; timeout 1500 python -m pytest tests/test_hip_distributed.py tests/test_hip_multirank.py -x -q > gpurun_out/r06_call11_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/r06_call11_tests.log | tail -3
echo "== bench (driver arguments)"; timeout 1200 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06_call11_bench_driver_args.json; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_call11_bench_driver_args.json").read())
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel"], d.get("parity",{}).get("ok"), d.get("extra_error"))
sh=d["extra"]["slab_share_to_self"]
print({k:(v["with_exchange_ms_per_step"], v["without_exchange_ms_per_step"], v["projected_speedup"]) for k,v in sh.items() if isinstance(v,dict)})
PY
