#!/bin/bash
# round 5, call 16: DIRECT fast block loop: waves per interior sweep x priority of the halo stream
mkdir -p gpurun_out/r5c
cd /root/repo
L=gpurun_out/r5c/probe_block_direct_caps.log
: > $L
for prio in 0 1; do
for cap in 1792 1536 1280 1024; do
  echo "== sweeps of $cap waves, PDEHIP_HALO_PRIORITY=$prio" >> $L
  PROBE_ONLY=0 PDEHIP_HALO_PRIORITY=$prio PDEHIP_EULER2=4,$cap timeout 300 python tools/probe_block.py 256,128,512 400 2>&1 | grep "ms/step" >> $L
done
done
cat $L
