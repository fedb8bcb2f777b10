#!/bin/bash
# round 5, call 9: kernel timelines of the fast block loop, schedule 2 (rim on the halo stream), without / with 64 reserved compute units
mkdir -p gpurun_out/r5b
cd /root/repo
export TMPDIR=/tmp
for cus in 0 64; do
PROBE_ONLY=0 PDEHIP_BLOCK2_MODE=2 PDEHIP_BLOCK2_CUS=$cus rocprofv3 --kernel-trace -d gpurun_out/r5b/trace_m2_$cus -- python tools/probe_block.py 256,128,512 40 > gpurun_out/r5b/trace_m2_$cus.log 2>&1
python tools/rocprof_timeline.py gpurun_out/r5b/trace_m2_$cus 0 4000 > gpurun_out/r5b/tl_all.txt 2>&1
N=$(wc -l < gpurun_out/r5b/tl_all.txt)
python tools/rocprof_timeline.py gpurun_out/r5b/trace_m2_$cus $((N-40)) 36 > gpurun_out/r5b/timeline_mode2_cus$cus.txt 2>&1
rm -rf gpurun_out/r5b/trace_m2_$cus gpurun_out/r5b/tl_all.txt
done
cat gpurun_out/r5b/timeline_mode2_cus0.txt; echo; cat gpurun_out/r5b/timeline_mode2_cus64.txt
