#!/bin/bash
# round 5, call 8: schedules x compute-unit partitions of the fast block loop (halos to self), 256 x 128 x 512
mkdir -p gpurun_out/r5b
cd /root/repo
L=gpurun_out/r5b/probe_block_modes.log
: > $L
for mode in 0 1 2; do
for cus in 0 16 32 64; do
  echo "== mode $mode, $cus compute units for the halo stream" >> $L
  PROBE_ONLY=0 PDEHIP_BLOCK2_MODE=$mode PDEHIP_BLOCK2_CUS=$cus timeout 300 python tools/probe_block.py 256,128,512 400 2>&1 | grep "ms/step" >> $L
done
done
PROBE_ONLY=1 timeout 300 python tools/probe_block.py 256,128,512 400 2>&1 | grep "ms/step" >> $L
cat $L
for mode in 1 2; do
PDEHIP_BLOCK2_MODE=$mode PDEHIP_BLOCK2_CUS=32 timeout 600 python -m pytest tests/test_hip_distributed.py -m gpu -x -q -k "fast_block_loop" 2>&1 | grep -E "passed|failed"
done
