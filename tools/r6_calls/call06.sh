#!/bin/bash
# round 6, call 6: tools/e2_bench6 with ping-pong launches (the time loop's access pattern) - streaming / plain stores, early / late loads
mkdir -p gpurun_out
for n in 256 384 512; do echo "=== $n ping-pong"; timeout 300 ./tools/e2_bench6 $n 20 2 1; done 2>&1 | tee gpurun_out/r06_call06_e2_bench6_pingpong.log
