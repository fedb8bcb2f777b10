#!/bin/bash
mkdir -p gpurun_out
{
for cfg in 4,4096,4 4,4096,22 4,4096,41 4,4096,21 4,4096,12 4,4096,1 4,2048,22 4,8192,22; do PDEHIP_EULER2=$cfg timeout 100 python tools/time_euler2.py 512 200 2>&1 | tail -1; done
} | tee gpurun_out/time_wg_shape.log
timeout 600 python -m pytest tests/test_hip_euler2.py -x -q 2>&1 | tail -1
