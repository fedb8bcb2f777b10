#!/bin/bash
# round 4, call 18: the strip of split rows with one cell per thread (rows in neighbouring lanes), by tail length
O=gpurun_out/r4r
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_tails.py tests/test_hip_operators.py tests/test_hip_derivatives.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/tests.log 2>&1
echo "tests rc=$?"; tail -2 $O/tests.log
for sp in 8 2 0; do
  echo "== PDEHIP_ROW_SPLIT=$sp"
  PDEHIP_ROW_SPLIT=$sp timeout 300 python tools/time_sizes.py 513x513x513 512x512x514 512x512x516 512x512x520 512x512x512 2>/dev/null | grep "^| 5" | cut -c1-60
done | tee $O/split_by_tail.log
