#!/bin/bash
mkdir -p gpurun_out
{
for cfg in default 4,1024 4,2048 4,512; do
  if [ $cfg = default ]; then timeout 200 python tools/probe_slab.py 64,512,512 300 2>&1 | grep "exchange=True"; else PDEHIP_EULER2=$cfg timeout 200 python tools/probe_slab.py 64,512,512 300 2>&1 | grep "exchange=True" | sed "s/^/[$cfg] /"; fi
done
for ch in 4 8 16; do NCCL_NCHANNELS_PER_PEER=$ch timeout 200 python tools/probe_slab.py 64,512,512 300 2>&1 | grep "exchange=True" | sed "s/^/[NCHANNELS_PER_PEER=$ch] /"; done
} | tee gpurun_out/probe_slab3.log
