#!/bin/bash
# round 4, call 3: merged slab sweep with the count in device memory (one store to the signal cell per sweep)
O=gpurun_out/r4d
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_distributed.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/gpu_slab.log 2>&1
echo "rc=$?"; tail -3 $O/gpu_slab.log; grep "^FAILED\|^ERROR" $O/gpu_slab.log | head
for m in 1 0 1 0; do
  echo "-- PDEHIP_SLAB_MERGED=$m"
  PDEHIP_SLAB_MERGED=$m timeout 120 python tools/probe_slab.py 64,512,512 400 2>&1 | grep "slab stepper exchange=True"
done | tee $O/probe_ab.log
for w in 1536 2560 3072 4096; do
  echo "-- merged, PDEHIP_MERGED_WAVES=$w"
  PDEHIP_MERGED_WAVES=$w timeout 120 python tools/probe_slab.py 64,512,512 400 2>&1 | grep "slab stepper exchange=True"
done | tee -a $O/probe_ab.log
timeout 120 python tools/probe_slab.py 64,512,512 400 2>&1 | grep "slab\|euler_run" | tee -a $O/probe_ab.log
timeout 120 python tools/probe_slab.py 128,512,512 300 2>&1 | grep "slab stepper" | tee -a $O/probe_ab.log
timeout 120 python tools/probe_slab.py 32,512,512 400 2>&1 | grep "slab stepper" | tee -a $O/probe_ab.log
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o merged -- python $GRAFT_REPO_ROOT/tools/probe_slab.py 64,512,512 60 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1; cd $GRAFT_REPO_ROOT
