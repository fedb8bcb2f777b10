#!/bin/bash
# round 6, call 13: x-chunk counts of the two-step sweep at sizes whose tile count is just above a divisor of the wave slots (513^3: 516 tiles)
mkdir -p gpurun_out
{
for n in 513 515 500,500,300 514; do
  for cap in 0 2064 3096 4128 6192 8256 16512; do
    if [ $cap = 0 ]; then echo "-- $n default"; timeout 200 python tools/time_euler2.py $n 100 2>&1 | tail -1
    else echo "-- $n PDEHIP_EULER2=4,$cap"; PDEHIP_EULER2=4,$cap timeout 200 python tools/time_euler2.py $n 100 2>&1 | tail -1; fi
  done
done
} | tee gpurun_out/r06_call13_chunks.log
