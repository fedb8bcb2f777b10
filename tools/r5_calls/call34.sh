#!/bin/bash
# round 5, call 34: schedule 2 of the fast block loop with the interior sweep of a pair released one stream hand-over behind the dispatch of the
# pair's rim kernel (PDEHIP_BLOCK2_GO, default on; 0 = before) - parity, A/B of ms per step (two / three cut axes), timeline
mkdir -p gpurun_out/r5o
cd /root/repo
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_hip_distributed.py -m gpu -x -q -k "fast_block or block_layer" 2>&1 | grep -E "passed|failed"
L=gpurun_out/r5o/probe_go.log
: > $L
for rep in 1 2 3; do
for go in 1 0; do
  echo "== PDEHIP_BLOCK2_GO=$go" >> $L
  PROBE_ONLY=0 PDEHIP_BLOCK2_GO=$go timeout 300 python tools/probe_block.py 256,128,512 400 2>&1 | grep "ms/step" >> $L
  PROBE_ONLY=0 PDEHIP_BLOCK2_GO=$go timeout 300 python tools/probe_block.py 256,256,256 400 2>&1 | grep "ms/step" >> $L
  PROBE_ONLY=0 PDEHIP_BLOCK2_GO=$go PDEHIP_PROBE_CUT_FASTEST=1 timeout 300 python tools/probe_block.py 256,256,256 400 2>&1 | grep "ms/step" >> $L
done
done
PROBE_ONLY=1 timeout 300 python tools/probe_block.py 256,128,512 400 2>&1 | grep "ms/step" >> $L
cat $L
cd /tmp
PROBE_ONLY=0 timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r5o/trace_go -- python $R/tools/probe_block.py 256,128,512 40 > /dev/null 2>&1
cd $R
TIMELINE_SKIP="fillBuffer|copyBuffer" python tools/rocprof_timeline.py gpurun_out/r5o/trace_go 100 14 | cut -c1-150 | tee gpurun_out/r5o/timeline_go.txt
find gpurun_out/r5o -name "*.db" -size +8M -delete
