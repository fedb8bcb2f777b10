#!/bin/bash
# round 5, call 1: access-order experiments for the Laplacian (tools/microbench6.hip) next to the product kernel on the same box
mkdir -p gpurun_out/r5a
cd /root/repo
timeout 300 tools/microbench6 512 > gpurun_out/r5a/microbench6_512.log 2>&1
python tools/time_lap.py 512 > gpurun_out/r5a/time_lap.log 2>&1
for t in "2,4,1,1,512" "2,4,1,1,2048" "2,2,1,1,1024" "2,2,1,1,2048" "2,1,1,1,1024" "2,1,1,1,2048" "2,1,1,1,4096"; do
  echo "TUNE $t" >> gpurun_out/r5a/time_lap.log
  PDEHIP_TUNE=$t timeout 120 python tools/time_lap.py 512 >> gpurun_out/r5a/time_lap.log 2>&1
done
tail -5 gpurun_out/r5a/time_lap.log
