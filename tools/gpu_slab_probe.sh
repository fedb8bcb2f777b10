#!/bin/bash
mkdir -p gpurun_out
echo "== euler2 + distributed tests"; timeout 900 python -m pytest tests/test_hip_euler2.py tests/test_hip_distributed.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -15
echo "== slab probe"
for s in 64,512,512 128,512,512 256,512,512; do timeout 300 python tools/probe_slab.py $s 300 2>&1 | grep "slab stepper\|euler_run\|single" ; done | tee gpurun_out/probe_slab2.log
echo "== slab probe, one-step loop"
PDEHIP_EULER2=off timeout 300 python tools/probe_slab.py 64,512,512 300 2>&1 | grep "slab stepper\|euler_run\|single" | tee -a gpurun_out/probe_slab2.log
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_r01.json
