#!/bin/bash
# round 3, call 6: reordered two-step slab loop (boundary + exchange enqueued first, priority halo stream)
O=gpurun_out/r3f
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_distributed.py tests/test_hip_multirank.py -m gpu -q --tb=short -p no:cacheprovider > $O/dist_pytest.log 2>&1
echo "rc=$?"; grep "passed\|failed" $O/dist_pytest.log | tail -1
for r in 1 2; do
  for p in 1 0; do
    echo "-- PDEHIP_HALO_PRIORITY=$p"
    PDEHIP_HALO_PRIORITY=$p timeout 300 python tools/probe_slab.py 64,512,512 400 2>&1 | grep "exchange=True"
  done
done | tee $O/probe_slab.log
timeout 300 python tools/probe_slab.py 64,512,512 400 2>&1 | grep "exchange=False\|euler_run" | tee -a $O/probe_slab.log
timeout 300 python tools/probe_slab.py 128,512,512 300 2>&1 | grep "exchange=" | tee -a $O/probe_slab.log
timeout 300 python bench.py --force-distributed --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300 | tee -a $O/probe_slab.log
