#!/bin/bash
# round 5, call 10: a SLAB through the fast block loop (only the first axis cut: rim kernel instead of the two-ended boundary sweep) against the slab loop
mkdir -p gpurun_out/r5b
cd /root/repo
L=gpurun_out/r5b/probe_slab_via_block2.log
: > $L
for shp in 64,512,512 128,512,512; do
for mode in 2 1 0; do
  echo "== block loop, cut 1,0,0, schedule $mode" >> $L
  PROBE_ONLY=0 PROBE_CUT=1,0,0 PDEHIP_BLOCK2_MODE=$mode timeout 300 python tools/probe_block.py $shp 400 2>&1 | grep "ms/step" >> $L
done
echo "== slab loop" >> $L
timeout 300 python tools/probe_slab.py $shp 400 2>&1 | grep "slab stepper\|euler_run\|single" >> $L
done
cat $L
