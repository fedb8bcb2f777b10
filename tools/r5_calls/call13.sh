#!/bin/bash
# round 5, call 13: cuts of the fastest axis in the fast block loop (transposed rim kernel), the rim kernel as boundary sweep of the slab loop
mkdir -p gpurun_out/r5c
cd /root/repo
timeout 900 python -m pytest tests/test_hip_distributed.py tests/test_hip_euler2.py -m gpu -x -q > gpurun_out/r5c/pytest_zcut_slabrim.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r5c/pytest_zcut_slabrim.log | tail -5
L=gpurun_out/r5c/probe_slab_rim.log
: > $L
for shp in 64,512,512 128,512,512; do
for rim in 1 0; do
  echo "== slab loop, PDEHIP_SLAB_RIM=$rim" >> $L
  PDEHIP_SLAB_RIM=$rim timeout 300 python tools/probe_slab.py $shp 400 2>&1 | grep "slab stepper" >> $L
done
done
echo "== block loop 256^3, all three axes cut (2 x 2 x 2 share)" >> $L
PROBE_ONLY=0 PDEHIP_PROBE_CUT_FASTEST=1 timeout 300 python tools/probe_block.py 256,256,256 400 2>&1 | grep "ms/step" >> $L
PROBE_ONLY=0 timeout 300 python tools/probe_block.py 256,256,256 400 2>&1 | grep "ms/step" >> $L
PROBE_ONLY=1 timeout 300 python tools/probe_block.py 256,256,256 400 2>&1 | grep "ms/step" >> $L
cat $L
