#!/bin/bash
# round 5, call 14: DIRECT mode of the fast block loop (rim kernel reads the receive buffer / writes the send buffer: no pack, no unpack)
mkdir -p gpurun_out/r5c
cd /root/repo
for mode in 2 1; do
PDEHIP_BLOCK2_MODE=$mode timeout 900 python -m pytest tests/test_hip_distributed.py -m gpu -x -q -k "fast_block_loop or block_layer_self" 2>&1 | grep -E "passed|failed|Error" | tail -3
done
L=gpurun_out/r5c/probe_block_direct.log
: > $L
for shp in 256,128,512 256,256,256; do
for d in 1 0; do
  echo "== schedule 2, PDEHIP_BLOCK2_DIRECT=$d" >> $L
  PROBE_ONLY=0 PDEHIP_BLOCK2_DIRECT=$d timeout 300 python tools/probe_block.py $shp 400 2>&1 | grep "ms/step" >> $L
done
echo "== schedule 1, direct" >> $L
PROBE_ONLY=0 PDEHIP_BLOCK2_MODE=1 timeout 300 python tools/probe_block.py $shp 400 2>&1 | grep "ms/step" >> $L
PROBE_ONLY=1 timeout 300 python tools/probe_block.py $shp 400 2>&1 | grep "ms/step" >> $L
done
echo "== 64 x 512 x 512 through the block loop, cut 1,0,0 (a slab)" >> $L
PROBE_ONLY=0 PROBE_CUT=1,0,0 timeout 300 python tools/probe_block.py 64,512,512 400 2>&1 | grep "ms/step" >> $L
cat $L
