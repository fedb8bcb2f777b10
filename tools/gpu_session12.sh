#!/bin/bash
mkdir -p gpurun_out
{
for r in 1 2; do timeout 100 python tools/time_euler2.py 512 200 2>&1 | tail -1; done
timeout 100 python tools/time_euler2.py 256 400 2>&1 | tail -1
timeout 100 python tools/time_euler2.py 512 200 float32 2>&1 | tail -1
timeout 100 python tools/time_euler2.py 4096,4096 400 2>&1 | tail -1
timeout 100 python tools/time_ch.py 512 40 2>&1 | tail -2
} | tee gpurun_out/time_load_order.log
timeout 900 python -m pytest tests/test_hip_euler2.py -x -q 2>&1 | tail -2
