#!/bin/bash
mkdir -p gpurun_out
{
for cfg in default 1,4096,2 1,4096,4 1,4096,8 1,1024,4 off; do
  for sh in 64,64 1024,1024 4096,4096; do
    if [ $cfg = default ]; then timeout 100 python tools/time_euler2.py $sh 400 2>&1 | tail -1; else PDEHIP_EULER2=$cfg timeout 100 python tools/time_euler2.py $sh 400 2>&1 | tail -1; fi
  done
done
for cfg in default 1,4096,4 off; do
  if [ $cfg = default ]; then timeout 100 python tools/time_ch.py 512,512 400 2>&1 | tail -2; else PDEHIP_EULER2=$cfg timeout 100 python tools/time_ch.py 512,512 400 2>&1 | tail -2; fi
done
} | tee gpurun_out/time_2d.log
python tools/bench_configs.py 2>&1 | tail -8 | head -4
