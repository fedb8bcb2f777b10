#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
for s in 64 256; do
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/trace_slab_$s -- python $R/tools/probe_slab.py $s,512,512 60 > /dev/null 2>&1
done
