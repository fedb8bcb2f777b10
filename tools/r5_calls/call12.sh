#!/bin/bash
# round 5, call 12: tile / chunk / workgroup-shape knobs of the two-step sweep and the Laplacian again, now that rows sit on 128-byte lines
mkdir -p gpurun_out/r5c
cd /root/repo
L=gpurun_out/r5c/knobs_after_alignment.log
: > $L
for e in default 4,2048 4,4096 4,3072 4,1024 8 4,2048,1 4,2048,2 4,2048,4 4,2048,21 4,2048,22 2,2048 2,4096; do
  echo -n "PDEHIP_EULER2=$e : " >> $L
  if [ $e = default ]; then python tools/time_euler2.py 512 400 2>/dev/null | grep EULER2 >> $L; else PDEHIP_EULER2=$e python tools/time_euler2.py 512 400 2>/dev/null | grep EULER2 >> $L; fi
done
for t in default 2,4,1,1,512 2,4,1,1,1024 2,4,1,1,2048 2,2,1,1,1024 2,2,1,1,2048 4,4,1,1,512 4,4,1,1,1024 4,2,1,1,1024; do
  echo -n "PDEHIP_TUNE=$t : " >> $L
  if [ $t = default ]; then python tools/time_lap.py 512 2>/dev/null | grep LAP >> $L; else PDEHIP_TUNE=$t python tools/time_lap.py 512 2>/dev/null | grep LAP >> $L; fi
done
cat $L
