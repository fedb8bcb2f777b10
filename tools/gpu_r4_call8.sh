#!/bin/bash
# round 4, call 8: row split of the one-step kernels (513^3, 511^3, 512x512x513), tails tests, size table
O=gpurun_out/r4h
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_tails.py tests/test_hip_operators.py tests/test_hip_derivatives.py tests/test_hip_steppers.py -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_tails.log 2>&1
echo "rc=$?"; tail -3 $O/gpu_tails.log; grep "^FAILED\|^ERROR" $O/gpu_tails.log | head
for split in 8 0; do
  echo "-- PDEHIP_ROW_SPLIT=$split"
  PDEHIP_ROW_SPLIT=$split timeout 300 python tools/time_sizes.py 513x513x513 511x511x511 512x512x513 512x512x512 510x510x510 4095x4097 2>&1 | grep "^|"
done | tee $O/time_sizes.log
