#!/bin/bash
# round 4, call 2: merged slab sweep (boundary layers first + hipStreamWaitValue64) - correctness, then the self-exchange probe A/B
O=gpurun_out/r4b
mkdir -p $O
export TMPDIR=/tmp
echo "== correctness: slab loops with the exchange to self (merged sweep is the default)"
timeout 600 python -m pytest tests/test_hip_distributed.py tests/test_hip_euler2.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/gpu_slab.log 2>&1
echo "rc=$?"; tail -3 $O/gpu_slab.log; grep "^FAILED\|^ERROR" $O/gpu_slab.log | head
echo "== probe 64x512x512"
for m in 1 0 1 0; do
  echo "-- PDEHIP_SLAB_MERGED=$m"
  PDEHIP_SLAB_MERGED=$m timeout 120 python tools/probe_slab.py 64,512,512 400 2>&1 | grep "slab stepper exchange=True"
done | tee $O/probe_ab.log
for w in 1536 2560 3072 4096; do
  echo "-- merged, PDEHIP_MERGED_WAVES=$w"
  PDEHIP_MERGED_WAVES=$w timeout 120 python tools/probe_slab.py 64,512,512 400 2>&1 | grep "slab stepper exchange=True"
done | tee -a $O/probe_ab.log
timeout 120 python tools/probe_slab.py 64,512,512 400 2>&1 | tee -a $O/probe_ab.log
timeout 120 python tools/probe_slab.py 128,512,512 300 2>&1 | grep "slab stepper" | tee -a $O/probe_ab.log
timeout 120 python tools/probe_slab.py 32,512,512 400 2>&1 | grep "slab stepper" | tee -a $O/probe_ab.log
echo "== plain two-step kernel (regression check of the shared kernel body)"
timeout 120 python tools/time_euler2.py 512 200 2>&1 | tail -2 | tee $O/time_e2.log
echo "== kernel trace of the merged loop"
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o merged -- python $GRAFT_REPO_ROOT/tools/probe_slab.py 64,512,512 60 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1; cd $GRAFT_REPO_ROOT
ls $O/trace* | head
