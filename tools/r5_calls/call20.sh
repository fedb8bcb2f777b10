#!/bin/bash
# round 5, call 20: two planes of prefetch in the one-step kernel (PF = 2) against the default (PF = 1), 512^3 fp64 Laplacian, alternating
mkdir -p gpurun_out/r5d
cd /root/repo
L=gpurun_out/r5d/lap_prefetch.log
: > $L
for rep in 1 2 3; do
for t in default 2,4,1,2,1024 2,4,1,2,2048 2,4,1,2,512 2,2,1,2,1024 2,2,1,2,2048 2,2,1,2,3072; do
  echo -n "PDEHIP_TUNE=$t : " >> $L
  if [ $t = default ]; then python tools/time_lap.py 512 2>/dev/null | grep LAP >> $L; else PDEHIP_TUNE=$t python tools/time_lap.py 512 2>/dev/null | grep LAP >> $L; fi
done
done
cat $L
