#!/bin/bash
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -4
echo "== slab probe"
for s in 64,512,512 128,512,512 256,512,512; do timeout 300 python tools/probe_slab.py $s 300 2>&1 | grep "exchange=True\|euler_run" ; done | tee gpurun_out/probe_slab2.log
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_r01.json
echo "== bench forced distributed (1 rank, halo to self)"; timeout 600 python bench.py --force-distributed 2>&1 | tail -1 | tee gpurun_out/bench_r01_forced.json
