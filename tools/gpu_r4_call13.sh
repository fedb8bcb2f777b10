#!/bin/bash
# round 4, call 13: open rows of the two-step sweep (tiles over whole chunks + the remaining columns by the shell kernel)
O=gpurun_out/r4m
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_euler2.py tests/test_hip_tails.py tests/test_hip_steppers.py tests/test_hip_frows.py tests/test_hip_properties.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/gpu_tests.log 2>&1
echo "rc=$?"; tail -6 $O/gpu_tests.log
echo "== open rows"; timeout 300 python tools/time_sizes.py 513x513x513 512x512x513 514x514x514 512x512x512 2>/dev/null | tee $O/sizes_open.log | grep "^| 5" | cut -c1-110
echo "== closed rows"; PDEHIP_OPEN_ROWS=0 timeout 300 python tools/time_sizes.py 513x513x513 512x512x513 514x514x514 2>/dev/null | tee $O/sizes_closed.log | grep "^| 5" | cut -c1-110
