#!/bin/bash
# round 5, call 19: why does cfg2 (1024^2, 1000 Euler steps through eq.solve) take 43 us per step in `bench.py --steps 20 --warmup 5` and 5 us with the defaults?
mkdir -p gpurun_out/r5d
cd /root/repo
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], "value", d["value"], "cfg2", d["extra"]["cfg2_diffusion_1024sq_f64_euler"], "tile2d", d["roofline_operators"]["tile2d"]["cfg2_diffusion_1024sq"]["us_per_step"])
PY
}
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5d/a.json; show gpurun_out/r5d/a.json "A steps 20 warmup 5:"
PDEHIP_GRAPH=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5d/b.json; show gpurun_out/r5d/b.json "B same, PDEHIP_GRAPH=0:"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5d/c.json; show gpurun_out/r5d/c.json "C steps 200 warmup 20:"
timeout 300 python bench.py --steps 20 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5d/d.json; show gpurun_out/r5d/d.json "D steps 20 warmup 20:"
timeout 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5d/e.json; show gpurun_out/r5d/e.json "E steps 200 warmup 5:"
