#!/bin/bash
# round 5, call 38: rim kernel limited to 128 vector registers (4 waves per SIMD; it spills) - does it get through faster next to a sweep?
mkdir -p gpurun_out/r5s
cd /root/repo
L=gpurun_out/r5s/probe_rim128.log
: > $L
timeout 600 python -m pytest tests/test_hip_distributed.py -m gpu -q -k "fast_block or two_steps_per_sweep" 2>&1 | grep -E "passed|failed" >> $L
for rep in 1 2; do
  timeout 300 python tools/probe_slab.py 64,512,512 400 2>&1 | grep "slab stepper" >> $L
  PROBE_ONLY=0 timeout 300 python tools/probe_block.py 256,128,512 400 2>&1 | grep "ms/step" >> $L
  PROBE_ONLY=0 PDEHIP_PROBE_CUT_FASTEST=1 timeout 300 python tools/probe_block.py 256,256,256 400 2>&1 | grep "ms/step" >> $L
done
PROBE_ONLY=1 timeout 300 python tools/probe_block.py 256,128,512 400 2>&1 | grep "ms/step" >> $L
cat $L
