#!/bin/bash
# round 4, call 16: after the fix of the image range in the shell kernel; row split of the one-step kernels by tail length
O=gpurun_out/r4p
mkdir -p $O
export TMPDIR=/tmp
for k in 1 2; do
  timeout 900 python -m pytest tests/test_hip_frows.py tests/test_hip_euler2.py tests/test_hip_complex.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/tests_$k.log 2>&1
  echo "run $k rc=$?"; tail -2 $O/tests_$k.log
done
for sp in 8 2 0; do
  echo "== PDEHIP_ROW_SPLIT=$sp"
  PDEHIP_ROW_SPLIT=$sp timeout 300 python tools/time_sizes.py 512x512x516 515x515x515 514x514x514 512x512x520 2>/dev/null | grep "^| 5" | cut -c1-60
done | tee $O/split_by_tail.log
