#!/bin/bash
# round 5, call 6: the fast block loop on the GPU - parity tests, then the self-exchange probe
mkdir -p gpurun_out/r5b
cd /root/repo
timeout 900 python -m pytest tests/test_hip_distributed.py -m gpu -x -q -k "fast_block_loop or block_layer_self" > gpurun_out/r5b/pytest_block2.log 2>&1
tail -15 gpurun_out/r5b/pytest_block2.log
for shp in 256,128,512 256,256,256 128,256,512; do
  timeout 300 python tools/probe_block.py $shp 400 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5b/probe_block.log
done
cat gpurun_out/r5b/probe_block.log
