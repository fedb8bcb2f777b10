#!/bin/bash
# round 5, call 4: interleaved A/B of the row alignment on the slab / cache-resident shapes (5 alternations each)
mkdir -p gpurun_out/r5a
cd /root/repo
L=gpurun_out/r5a/ab_row_align_slabs.log
: > $L
for shape in 64,512,512 128,512,512 256,256,256 32,256,256; do
for rep in 1 2 3 4 5; do
for al in 16 128; do
  echo -n "align=$al " >> $L
  PDEHIP_ROW_ALIGN=$al python tools/time_euler2.py $shape 1000 2>/dev/null | grep EULER2 >> $L
done
done
done
for rep in 1 2 3; do
for al in 16 128; do
  echo -n "align=$al " >> $L
  PDEHIP_ROW_ALIGN=$al python tools/time_euler2.py 256 1000 float32 2>/dev/null | grep EULER2 >> $L
done
done
cat $L
