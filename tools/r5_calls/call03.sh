#!/bin/bash
# round 5, call 3: A/B of the row alignment inside the product (PDEHIP_ROW_ALIGN=16 = rounds 1-4, 128 = new default) + the GPU suite
mkdir -p gpurun_out/r5a
cd /root/repo
L=gpurun_out/r5a/ab_row_align.log
: > $L
for rep in 1 2; do
for al in 16 128 256; do
  echo "== PDEHIP_ROW_ALIGN=$al" >> $L
  PDEHIP_ROW_ALIGN=$al python tools/time_lap.py 512 2>/dev/null | grep LAP >> $L
  PDEHIP_ROW_ALIGN=$al python tools/time_euler2.py 512 200 2>/dev/null | grep EULER2 >> $L
  PDEHIP_ROW_ALIGN=$al python tools/time_lap.py 512 float32 2>/dev/null | grep LAP >> $L
  PDEHIP_ROW_ALIGN=$al python tools/time_euler2.py 512 200 float32 2>/dev/null | grep EULER2 >> $L
  PDEHIP_ROW_ALIGN=$al python tools/time_euler2.py 64,512,512 400 2>/dev/null | grep EULER2 >> $L
  PDEHIP_ROW_ALIGN=$al python tools/time_euler2.py 256 400 2>/dev/null | grep EULER2 >> $L
done
done
cat $L
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5a/pytest_gpu_align128.log 2>&1
tail -5 gpurun_out/r5a/pytest_gpu_align128.log
