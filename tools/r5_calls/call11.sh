#!/bin/bash
# round 5, call 11: the interior sweep of the fast block loop with fewer than one full round of waves (room for the small kernels)
mkdir -p gpurun_out/r5b
cd /root/repo
L=gpurun_out/r5b/probe_block_cap.log
: > $L
for mode in 2 1; do
for cap in 2048 1792 1536 1280 1024; do
  echo "== schedule $mode, sweeps of $cap waves" >> $L
  PROBE_ONLY=0 PDEHIP_EULER2=4,$cap PDEHIP_BLOCK2_MODE=$mode timeout 300 python tools/probe_block.py 256,128,512 400 2>&1 | grep "ms/step" >> $L
done
done
for cap in 2048 1536; do
  echo "== no exchange, sweeps of $cap waves" >> $L
  PROBE_ONLY=1 PDEHIP_EULER2=4,$cap timeout 300 python tools/probe_block.py 256,128,512 400 2>&1 | grep "ms/step" >> $L
done
cat $L
