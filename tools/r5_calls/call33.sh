#!/bin/bash
# round 5, call 33: fast block loop with all three axes cut - waves per interior sweep (the interior sweep is not on the critical path there:
# does a smaller sweep leave the halo stream's launches more room?)
mkdir -p gpurun_out/r5n
cd /root/repo
L=gpurun_out/r5n/probe_zcut_caps.log
: > $L
for rep in 1 2; do
for cap in 1792 1280 1024 768 512; do
  echo "== sweeps of $cap waves" >> $L
  PROBE_ONLY=0 PDEHIP_PROBE_CUT_FASTEST=1 PDEHIP_EULER2=4,$cap timeout 300 python tools/probe_block.py 256,256,256 400 2>&1 | grep "ms/step" >> $L
done
done
cat $L
