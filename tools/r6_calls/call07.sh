#!/bin/bash
# round 6, call 7: 4-row fp64 tile with late loads + branches (faces), tall tile only for all-periodic grids - whole GPU suite, timings with
# faces / periodic at several sizes, bench line with the new extras (slab_share_to_self, kernel instance from the library, launch floor)
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== GPU suite"; timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
{
for rep in 1 2; do
  for n in 512 256 128,512,512 64,512,512; do
    echo "-- $n periodic"; timeout 200 python tools/time_euler2.py $n 200 2>&1 | tail -1
    echo "-- $n walls";    TIME_PERIODIC=0 timeout 200 python tools/time_euler2.py $n 200 2>&1 | tail -1
  done
done
} | tee gpurun_out/r06_call07_time_euler2.log
echo "== bench (driver arguments)"; timeout 1200 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r06_call07_bench_driver_args.json; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_call07_bench_driver_args.json").read())
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline_operator"]["frac"], d.get("parity"), d.get("extra_error"))
print(json.dumps(d["extra"]["slab_share_to_self"], indent=1))
print(json.dumps(d["roofline_operators"].get("tile2d"), indent=1)[:1200])
print(json.dumps(d["cpu_baseline"], indent=1)[:1500])
PY
