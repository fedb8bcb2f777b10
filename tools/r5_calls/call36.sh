#!/bin/bash
# round 5, call 36: kernel timeline of the DEFAULT slab loop (64 x 512 x 512, halos to self)
mkdir -p gpurun_out/r5q
cd /root/repo
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r5q/trace_slab -- python $R/tools/probe_slab.py 64,512,512 40 > /dev/null 2>&1
cd $R
TIMELINE_SKIP="fillBuffer|copyBuffer" python tools/rocprof_timeline.py gpurun_out/r5q/trace_slab 60 18 | cut -c1-150 | tee gpurun_out/r5q/timeline_slab.txt
find gpurun_out/r5q -name "*.db" -size +8M -delete
