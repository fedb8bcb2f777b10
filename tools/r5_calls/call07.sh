#!/bin/bash
# round 5, call 7: kernel timeline of the fast block loop (halos to self) at 256 x 128 x 512
mkdir -p gpurun_out/r5b
cd /root/repo
export TMPDIR=/tmp
PROBE_ONLY=0 rocprofv3 --kernel-trace -d gpurun_out/r5b/trace_fast -- python tools/probe_block.py 256,128,512 40 > gpurun_out/r5b/trace_fast.log 2>&1
python tools/rocprof_timeline.py gpurun_out/r5b/trace_fast 0 4000 > gpurun_out/r5b/timeline_all.txt 2>&1
N=$(wc -l < gpurun_out/r5b/timeline_all.txt)
python tools/rocprof_timeline.py gpurun_out/r5b/trace_fast $((N-90)) 90 > gpurun_out/r5b/timeline_fast.txt 2>&1
cat gpurun_out/r5b/timeline_fast.txt
rm -rf gpurun_out/r5b/trace_fast
