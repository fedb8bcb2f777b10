#!/bin/bash
# round 4: the real py-pde drives the slab / block solver (`solver="hip_slab"`) on the MI355X (one rank, RCCL to self) - the worker of
# tests/test_distributed_gloo.py::test_real_pypde_drives_the_slab_path with its differential fuzz, against the reference's serial runs
O=gpurun_out/r4dropin
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
export PDEHIP_REFERENCE=$R/_refscratch PDEHIP_DROPIN_REAL=1
for d in slab auto; do
  PDEHIP_WORKER_FUZZ=12 PDEHIP_WORKER_DECOMPOSITION=$d timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29611 tests/pypde_slab_worker.py > $O/slab_worker_$d.log 2>&1
  echo "worker $d rc=$?"; grep PYPDESLAB $O/slab_worker_$d.log | tail -1 | cut -c1-300; grep -o '"failures": \[[^]]*\]' $O/slab_worker_$d.log | tail -1
done
