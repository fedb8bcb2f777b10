#!/bin/bash
# round 6, call 14: slab schedule 4 (second boundary pass behind A_int of its group: 4 + 4 instead of 8 + 4 boundary layers per side) against schedule 3
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_distributed.py -x -q -k "four_steps" > gpurun_out/r06_call14_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/r06_call14_tests.log | tail -2
probe() { timeout 300 python tools/probe_slab.py "$@" 2>&1 | grep "slab stepper exchange=True"; }
{
for rep in 1 2 3; do
  for s in 64,512,512 128,512,512; do
    for m in 3 4; do echo "-- $s schedule $m"; PDEHIP_SLAB_DEEP_MODE=$m probe $s 400; done
  done
done
} | tee gpurun_out/r06_call14_schedule4.log
cd /tmp && PDEHIP_SLAB_DEEP_MODE=4 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06_call14_trace -o slab -- python $GRAFT_REPO_ROOT/tools/probe_slab.py 64,512,512 40 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
TIMELINE_SKIP="fillBuffer|copyBuffer|spin" python tools/rocprof_timeline.py gpurun_out/r06_call14_trace 30 40 2>&1 | tee gpurun_out/r06_call14_timeline.txt
