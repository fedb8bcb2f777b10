#!/bin/bash
# round 3, call 27: after making the zero fill of pdehip_malloc synchronous: race probe, slab worker fuzz x4, the suites that allocate most
O=gpurun_out/r3j
mkdir -p $O
export TMPDIR=/tmp
timeout 120 ./tools/malloc_race 20000 2>&1 | tee $O/malloc_race.log | grep MALLOCRACE
R=$PWD
for i in 1 2 3 4; do
  PDEHIP_REFERENCE=$R/_refscratch PDEHIP_WORKER_FUZZ=18 PDEHIP_WORKER_DECOMPOSITION=slab timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 2961$i tests/pypde_slab_worker.py > $O/slab_worker_$i.log 2>&1
  echo "worker $i rc=$?"; grep -o '"failures": \[[^]]*\]' $O/slab_worker_$i.log | tail -1
done
timeout 900 python -m pytest tests/test_hip_distributed.py tests/test_hip_frows.py tests/test_hip_steppers.py tests/test_transfers.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_subset.log 2>&1
echo "rc=$?"; tail -1 $O/pytest_subset.log
