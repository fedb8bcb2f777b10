#!/bin/bash
# round 5, call 37: the default slab loop (64 x 512 x 512, halos to self) with other numbers of wave tiles per sweep (one round of waves = 2048;
# more = a second round, at whose start the boundary kernel could get in)
mkdir -p gpurun_out/r5r
cd /root/repo
L=gpurun_out/r5r/probe_slab_tiles.log
: > $L
for rep in 1 2; do
for cap in 1536 1024 2048 3072 4096 6144; do
  echo "== sweeps of $cap wave tiles" >> $L
  PDEHIP_EULER2=4,$cap timeout 300 python tools/probe_slab.py 64,512,512 400 2>&1 | grep "slab stepper" >> $L
done
done
cat $L
