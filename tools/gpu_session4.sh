#!/bin/bash
mkdir -p gpurun_out
echo "== euler2 tests"; timeout 900 python -m pytest tests/test_hip_euler2.py -x -q 2>&1 | tail -15
echo "== timing"
for cfg in off 4,4096 4,2048 4,8192 2,4096 2,8192 4,1024; do PDEHIP_EULER2=$cfg timeout 120 python tools/time_euler2.py 512 200 2>&1 | tail -1; done | tee gpurun_out/time_euler2.log
