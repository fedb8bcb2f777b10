#!/bin/bash
# round 5, call 5: A/B of the tight row pitch (n2 + 2 rounded up to 128 B instead of lpad + n2 + 1): footprint vs the Infinity Cache
mkdir -p gpurun_out/r5a
cd /root/repo
L=gpurun_out/r5a/ab_row_pitch.log
: > $L
for shape in 512,512,512 64,512,512 128,512,512 256,256,256; do
for rep in 1 2 3; do
for pt in loose tight; do
  echo -n "pitch=$pt " >> $L
  PDEHIP_ROW_PITCH=$pt python tools/time_euler2.py $shape 600 2>/dev/null | grep EULER2 >> $L
done
done
done
for rep in 1 2 3; do
for pt in loose tight; do
  echo -n "pitch=$pt " >> $L
  PDEHIP_ROW_PITCH=$pt python tools/time_lap.py 512 2>/dev/null | grep LAP >> $L
done
done
cat $L
PDEHIP_ROW_PITCH=tight timeout 900 python -m pytest tests/test_hip_euler2.py tests/test_hip_tails.py tests/test_hip_operators.py tests/test_hip_steppers.py tests/test_transfers.py tests/test_hip_distributed.py -m gpu -x -q > gpurun_out/r5a/pytest_gpu_tight.log 2>&1
tail -3 gpurun_out/r5a/pytest_gpu_tight.log
