#!/bin/bash
# round 6, call 2: schedule 3 of the four-steps-per-exchange slab loop (boundary layers a group ahead on the halo stream) - parity, ms per step
# to self against schedules 1 / 2 and the two-step loop, kernel timeline
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== slab tests"; timeout 1500 python -m pytest tests/test_hip_distributed.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -8
probe() { timeout 300 python tools/probe_slab.py "$@" 2>&1 | grep "slab stepper"; }
{
for s in 64,512,512 128,512,512 256,512,512; do
  for rep in 1 2; do
    for m in 3 2 1; do echo "-- $s four steps per exchange, schedule $m"; PDEHIP_SLAB_DEEP_MODE=$m probe $s 400; done
    echo "-- $s two steps per exchange (round 5)"; PDEHIP_SLAB_EULER4=0 probe $s 400
  done
done
} | tee gpurun_out/r06_call02_probe_slab.log
echo "== kernel timeline, 64 x 512 x 512, schedule 3"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06_call02_trace -o slab -- python $GRAFT_REPO_ROOT/tools/probe_slab.py 64,512,512 40 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
TIMELINE_SKIP="fillBuffer|copyBuffer" python tools/rocprof_timeline.py gpurun_out/r06_call02_trace 0 100 2>&1 | tee gpurun_out/r06_call02_timeline.txt
