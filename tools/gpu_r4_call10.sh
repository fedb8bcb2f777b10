#!/bin/bash
# round 4, call 10: strip workgroups inside the main launch (split rows); conditions of time and position on the two-step sweep
O=gpurun_out/r4j
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_tails.py tests/test_hip_operators.py tests/test_hip_frows.py tests/test_hip_steppers.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/gpu_tests.log 2>&1
echo "rc=$?"; tail -15 $O/gpu_tests.log
echo "== fused strip"; timeout 300 python tools/time_sizes.py 513x513x513 512x512x513 511x511x511 512x512x512 2>/dev/null | tee $O/sizes_fused.log | grep "^|" | cut -c1-110
echo "== separate strip"; PDEHIP_ROW_SPLIT_SEPARATE=1 timeout 300 python tools/time_sizes.py 513x513x513 512x512x513 2>/dev/null | tee $O/sizes_separate.log | grep "^| 5" | cut -c1-110
echo "== no split"; PDEHIP_ROW_SPLIT=0 timeout 300 python tools/time_sizes.py 513x513x513 512x512x512 2>/dev/null | tee $O/sizes_nosplit.log | grep "^| 5" | cut -c1-110
echo "== bc program"; timeout 600 python tools/time_bc_program.py 512 100 2>/dev/null | tee $O/bcprog_two.log | grep BCPROG
PDEHIP_TIMED_TWO_STEP=0 timeout 600 python tools/time_bc_program.py 512 100 2>/dev/null | tee $O/bcprog_one.log | grep BCPROG
