#!/bin/bash
# round 5, call 15: kernel timeline of the fast block loop, schedule 2, DIRECT
mkdir -p gpurun_out/r5c
cd /root/repo
export TMPDIR=/tmp
PROBE_ONLY=0 rocprofv3 --kernel-trace -d gpurun_out/r5c/trace_direct -- python tools/probe_block.py 256,128,512 40 > gpurun_out/r5c/trace_direct.log 2>&1
python tools/rocprof_timeline.py gpurun_out/r5c/trace_direct 0 4000 > gpurun_out/r5c/tl_all.txt 2>&1
N=$(wc -l < gpurun_out/r5c/tl_all.txt)
python tools/rocprof_timeline.py gpurun_out/r5c/trace_direct $((N-34)) 30 > gpurun_out/r5c/timeline_direct.txt 2>&1
rm -rf gpurun_out/r5c/trace_direct gpurun_out/r5c/tl_all.txt
cat gpurun_out/r5c/timeline_direct.txt
