#!/bin/bash
# round 5, call 39: the slab loop with 2-row tiles of the two-step sweep (fewer registers per wave: room for the boundary kernel's waves on every SIMD?)
mkdir -p gpurun_out/r5t
cd /root/repo
L=gpurun_out/r5t/probe_slab_ry2.log
: > $L
for rep in 1 2; do
for t in 4 2 2,2048 2,3072 2,1536; do
  echo "== PDEHIP_EULER2=$t" >> $L
  PDEHIP_EULER2=$t timeout 300 python tools/probe_slab.py 64,512,512 400 2>&1 | grep "slab stepper" >> $L
done
done
cat $L
