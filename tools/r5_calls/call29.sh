#!/bin/bash
# round 5, call 29: kernel timeline of the fast block loop with ALL THREE axes cut (the 2 x 2 x 2 share of 512^3: 256^3, halos to self)
mkdir -p gpurun_out/r5j
cd /root/repo
export TMPDIR=/tmp
R=$PWD
PROBE_ONLY=0 PDEHIP_PROBE_CUT_FASTEST=1 timeout 300 python tools/probe_block.py 256,256,256 400 2>&1 | grep "ms/step" | tee gpurun_out/r5j/probe_zcut.log
cd /tmp
PROBE_ONLY=0 PDEHIP_PROBE_CUT_FASTEST=1 timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r5j/trace_zcut -- python $R/tools/probe_block.py 256,256,256 40 > /dev/null 2>&1
cd $R
TIMELINE_SKIP="fillBuffer|copyBuffer" python tools/rocprof_timeline.py gpurun_out/r5j/trace_zcut 120 40 | cut -c1-150 | tee gpurun_out/r5j/timeline_zcut.txt
find gpurun_out/r5j -name "*.db" -size +8M -delete
