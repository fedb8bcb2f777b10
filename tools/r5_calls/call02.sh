#!/bin/bash
# round 5, call 2: does the 128-byte alignment of the rows matter? (tools/microbench6 with the product's pitch and with aligned rows)
mkdir -p gpurun_out/r5a
cd /root/repo
timeout 300 tools/microbench6 512 2 > gpurun_out/r5a/microbench6_512_pad2.log 2>&1
timeout 300 tools/microbench6 512 16 > gpurun_out/r5a/microbench6_512_pad16.log 2>&1
