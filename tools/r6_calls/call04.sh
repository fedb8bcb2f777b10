#!/bin/bash
# round 6, call 4: instruction-diet variants of the two-step sweep outside the library (tools/e2_bench6.hip), 512^3 and 256^3
mkdir -p gpurun_out
for n in 512 256; do echo "=== $n"; timeout 300 ./tools/e2_bench6 $n 20 3; done 2>&1 | tee gpurun_out/r06_call04_e2_bench6.log
