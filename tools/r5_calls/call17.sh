#!/bin/bash
# round 5, call 17: the tall tile of the two-step sweep (8 rows, one wave per SIMD, four plane buffers) against the default tile, alternating
mkdir -p gpurun_out/r5c
cd /root/repo
L=gpurun_out/r5c/ab_tall_tile.log
: > $L
for shape in 512,512,512 384,384,384 256,256,256 512,512,256 128,512,512; do
for rep in 1 2 3 4; do
  echo -n "default " >> $L; python tools/time_euler2.py $shape 400 2>/dev/null | grep EULER2 >> $L
  echo -n "tall    " >> $L; PDEHIP_EULER2=8 python tools/time_euler2.py $shape 400 2>/dev/null | grep EULER2 >> $L
done
done
cat $L
