#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_euler2.py -x -q 2>&1 | tail -2
for cfg in 4,4096,4 4,4096,1 4,4096,2 4,2048,4 4,8192,4 2,4096,4; do PDEHIP_EULER2=$cfg timeout 120 python tools/time_euler2.py 512 200 2>&1 | tail -1; done | tee gpurun_out/time_euler2_order.log
for cfg in 2,4096,1 2,4096,2; do PDEHIP_EULER2=$cfg timeout 120 python tools/time_euler2.py 512 200 float32 2>&1 | tail -1; done | tee -a gpurun_out/time_euler2_order.log
