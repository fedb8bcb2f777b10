#!/bin/bash
# round 5, call 32: the slab loop with its interior sweep released behind THIS pair's boundary sweep (PDEHIP_SLAB_GATE=1: the boundary kernel runs
# alone, the RCCL kernel is dispatched before the sweep fills the chip) against the default - parity, ms per step to self, timeline
mkdir -p gpurun_out/r5m
cd /root/repo
export TMPDIR=/tmp
R=$PWD
PDEHIP_SLAB_GATE=1 timeout 900 python -m pytest tests/test_hip_distributed.py -m gpu -x -q -k "two_steps_per_sweep or overlapped_self or both_physical" 2>&1 | grep -E "passed|failed"
L=gpurun_out/r5m/probe_slab_gate.log
: > $L
for rep in 1 2 3; do
for shp in 64,512,512 128,512,512; do
for gate in 0 1; do
  echo "== slab loop, PDEHIP_SLAB_GATE=$gate" >> $L
  PDEHIP_SLAB_GATE=$gate timeout 300 python tools/probe_slab.py $shp 400 2>&1 | grep "slab stepper exchange=True" >> $L
done
done
done
cat $L
cd /tmp
PDEHIP_SLAB_GATE=1 timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r5m/trace_gate -- python $R/tools/probe_slab.py 64,512,512 40 > /dev/null 2>&1
cd $R
TIMELINE_SKIP="fillBuffer|copyBuffer" python tools/rocprof_timeline.py gpurun_out/r5m/trace_gate 60 16 | cut -c1-150 | tee gpurun_out/r5m/timeline_gate.txt
find gpurun_out/r5m -name "*.db" -size +8M -delete
