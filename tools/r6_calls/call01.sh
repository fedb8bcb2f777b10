#!/bin/bash
# round 6, call 1: four steps per exchange on a slab (slab::euler4_run) - parity to self, then ms per step to self against the two-step loop
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== slab tests"; timeout 1200 python -m pytest tests/test_hip_distributed.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -8
probe() { timeout 300 python tools/probe_slab.py "$@" 2>&1 | grep "slab stepper\|euler_run\|single"; }
{
for s in 64,512,512 128,512,512 256,512,512; do
  for rep in 1 2; do
    echo "-- $s four steps per exchange, mode 1"; PDEHIP_SLAB_DEEP_MODE=1 probe $s 400
    echo "-- $s four steps per exchange, mode 2"; PDEHIP_SLAB_DEEP_MODE=2 probe $s 400
    echo "-- $s two steps per exchange (round 5)"; PDEHIP_SLAB_EULER4=0 probe $s 400
  done
done
} | tee gpurun_out/r06_call01_probe_slab.log
echo "== kernel timeline, 64 x 512 x 512, mode 1"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06_call01_trace -o slab -- python $GRAFT_REPO_ROOT/tools/probe_slab.py 64,512,512 40 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/r06_call01_trace | head
python tools/rocprof_timeline.py gpurun_out/r06_call01_trace 12 90 2>&1 | tee gpurun_out/r06_call01_timeline.txt
