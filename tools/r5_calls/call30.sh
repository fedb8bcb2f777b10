#!/bin/bash
# round 5, call 30: schedule 2 of the fast block loop with the next interior sweep released behind the PACK launch (cut fastest axis: the RCCL kernel
# no longer starves) - parity, ms per step for three / two cut axes, kernel timeline
mkdir -p gpurun_out/r5k
cd /root/repo
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_hip_distributed.py -m gpu -x -q -k "fast_block or block_layer" 2>&1 | tail -2
L=gpurun_out/r5k/probe_after_pack.log
: > $L
for rep in 1 2; do
PROBE_ONLY=0 PDEHIP_PROBE_CUT_FASTEST=1 timeout 300 python tools/probe_block.py 256,256,256 400 2>&1 | grep "ms/step" >> $L
PROBE_ONLY=0 timeout 300 python tools/probe_block.py 256,256,256 400 2>&1 | grep "ms/step" >> $L
PROBE_ONLY=0 timeout 300 python tools/probe_block.py 256,128,512 400 2>&1 | grep "ms/step" >> $L
done
PROBE_ONLY=1 timeout 300 python tools/probe_block.py 256,256,256 400 2>&1 | grep "ms/step" >> $L
cat $L
cd /tmp
PROBE_ONLY=0 PDEHIP_PROBE_CUT_FASTEST=1 timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r5k/trace_zcut -- python $R/tools/probe_block.py 256,256,256 40 > /dev/null 2>&1
cd $R
TIMELINE_SKIP="fillBuffer|copyBuffer" python tools/rocprof_timeline.py gpurun_out/r5k/trace_zcut 120 24 | cut -c1-150 | tee gpurun_out/r5k/timeline_zcut.txt
find gpurun_out/r5k -name "*.db" -size +8M -delete
