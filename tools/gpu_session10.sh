#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_euler2.py tests/test_hip_steppers.py -x -q 2>&1 | tail -4
for cfg in default off; do
  for args in "256 100 float64" "512 40 float64" "256 100 float32"; do
    if [ $cfg = off ]; then PDEHIP_EULER2=off timeout 200 python tools/time_ch.py $args 2>&1 | tail -2; else timeout 200 python tools/time_ch.py $args 2>&1 | tail -2; fi
  done
done | tee gpurun_out/time_ch.log
