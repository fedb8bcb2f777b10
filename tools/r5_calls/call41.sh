#!/bin/bash
# round 5, call 41: the two-step slab loop on streams with disjoint sets of compute units (PDEHIP_SLAB_CUS=<R> for the halo stream: the boundary kernel
# and the RCCL kernel get whole compute units instead of the half-empty SIMDs the capped sweep leaves) - parity, ms per step to self
mkdir -p gpurun_out/r5u
cd /root/repo
L=gpurun_out/r5u/probe_slab_cus.log
: > $L
PDEHIP_SLAB_CUS=64 timeout 600 python -m pytest tests/test_hip_distributed.py -m gpu -q -k "two_steps_per_sweep or overlapped_self" 2>&1 | grep -E "passed|failed" >> $L
for rep in 1 2; do
for cus in 0 64 48 32 96; do
  echo "== PDEHIP_SLAB_CUS=$cus" >> $L
  PDEHIP_SLAB_CUS=$cus timeout 300 python tools/probe_slab.py 64,512,512 400 2>&1 | grep "slab stepper exchange=True" >> $L
done
echo "== PDEHIP_SLAB_CUS=64, the first 64 of the numbering" >> $L
PDEHIP_SLAB_CUS=64 PDEHIP_SLAB_CUS_FIRST=1 timeout 300 python tools/probe_slab.py 64,512,512 400 2>&1 | grep "slab stepper exchange=True" >> $L
done
cat $L
