#!/bin/bash
# round 6, call 15: a 6-row tile (one wave per SIMD, 56 - 112 AGPRs) for the all-periodic two-step sweep, harness with ping-pong launches
mkdir -p gpurun_out
for n in 512 256 384; do echo "=== $n ping-pong"; timeout 300 ./tools/e2_bench6 $n 20 3 1; done 2>&1 | tee gpurun_out/r06_call15_six_rows.log
