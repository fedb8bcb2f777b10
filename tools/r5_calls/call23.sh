#!/bin/bash
# round 5, call 23: THICK boundary chunks of the slab loop on the compute stream (PDEHIP_SLAB_THICK layers per side; 0 = the thin boundary
# sweep on the halo stream) - parity tests, ms per step to self, kernel timeline
mkdir -p gpurun_out/r5f
cd /root/repo
timeout 900 python -m pytest tests/test_hip_distributed.py tests/test_hip_complex.py -m gpu -x -q > gpurun_out/r5f/pytest_thick.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r5f/pytest_thick.log | tail -5
L=gpurun_out/r5f/probe_slab_thick.log
: > $L
for rep in 1 2; do
for shp in 64,512,512 128,512,512; do
for thick in 0 8 4 12 16; do
  echo "== slab loop, PDEHIP_SLAB_THICK=$thick" >> $L
  PDEHIP_SLAB_THICK=$thick timeout 300 python tools/probe_slab.py $shp 400 2>&1 | grep "slab stepper exchange=True" >> $L
done
done
done
PDEHIP_SLAB_THICK=8 timeout 300 python tools/probe_slab.py 64,512,512 400 2>&1 | grep "slab stepper" >> $L
cat $L
export TMPDIR=/tmp
R=$PWD
cd /tmp
PDEHIP_SLAB_THICK=8 timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r5f/trace_thick -- python $R/tools/probe_slab.py 64,512,512 40 > /dev/null 2>&1
cd $R
python tools/rocprof_timeline.py gpurun_out/r5f/trace_thick 60 40 > gpurun_out/r5f/timeline_thick.txt 2>/dev/null
tail -24 gpurun_out/r5f/timeline_thick.txt
find gpurun_out/r5f -name "*.db" -size +8M -delete
