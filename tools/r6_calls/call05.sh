#!/bin/bash
# round 6, call 5: the all-periodic instances of the two-step sweep inside the library - parity (euler2 / baseline configs / properties), A/B
# against PDEHIP_E2_PER3=0 at several sizes, tall against 4-row, then the bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== parity"; timeout 1500 python -m pytest tests/test_hip_euler2.py tests/test_baseline_configs.py tests/test_hip_properties.py tests/test_hip_steppers.py -x -q 2>&1 | tail -4
{
for rep in 1 2 3; do
  for n in 512 256 384 128,512,512 512,512,256; do
    echo "-- $n default";        timeout 200 python tools/time_euler2.py $n 200 2>&1 | tail -1
    echo "-- $n PDEHIP_E2_PER3=0"; PDEHIP_E2_PER3=0 timeout 200 python tools/time_euler2.py $n 200 2>&1 | tail -1
    echo "-- $n 4-row tile";     PDEHIP_EULER2=4 timeout 200 python tools/time_euler2.py $n 200 2>&1 | tail -1
    echo "-- $n tall tile";      PDEHIP_EULER2=8 timeout 200 python tools/time_euler2.py $n 200 2>&1 | tail -1
  done
done
} | tee gpurun_out/r06_call05_per3_ab.log
echo "== bench (driver arguments)"; timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r06_call05_bench_driver_args.json; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_call05_bench_driver_args.json").read())
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline_operator"]["frac"], d.get("parity"))
PY
