#!/bin/bash
# round 6, call 3: the split library (three translation units of stencil kernels) - a quick suite subset; wave caps of the interior sweeps
# beside the boundary chain of slab schedule 3
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== tests (split translation units)"; timeout 1500 python -m pytest tests/test_hip_operators.py tests/test_hip_euler2.py tests/test_hip_tile2d.py tests/test_hip_steppers.py -x -q 2>&1 | tail -4
probe() { timeout 300 python tools/probe_slab.py "$@" 2>&1 | grep "slab stepper exchange=True"; }
{
for rep in 1 2; do
  for cap in 0 2048 1792 1280 1024; do
    echo "-- 64,512,512 schedule 3, PDEHIP_EULER2=4,$cap"
    if [ $cap = 0 ]; then probe 64,512,512 400; else PDEHIP_EULER2=4,$cap probe 64,512,512 400; fi
  done
done
} | tee gpurun_out/r06_call03_caps.log
