#!/bin/bash
# round 4, call 12: two-step sweep with conditions of time and position - tests and timings
O=gpurun_out/r4l
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_frows.py tests/test_hip_steppers.py tests/test_hip_tails.py tests/test_hip_euler2.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/gpu_tests.log 2>&1
echo "rc=$?"; tail -4 $O/gpu_tests.log
for k in 1 2 3; do timeout 600 python tools/time_bc_program.py 512 100 2>/dev/null | tee -a $O/bcprog_two.log | grep BCPROG; done
PDEHIP_TIMED_TWO_STEP=0 timeout 600 python tools/time_bc_program.py 512 100 2>/dev/null | tee $O/bcprog_one.log | grep BCPROG
timeout 300 python tools/time_sizes.py 513x513x513 511x511x511 512x512x513 512x512x512 500x500x300 2>/dev/null | tee $O/sizes.log | grep "^| 5" | cut -c1-110
echo "== 500x500x300 tile shapes (ry,cz,wy,pf,blocks)"
for tune in 2,1,1,1,4096 2,1,1,1,2048 2,2,1,1,1024 2,2,1,1,2048 2,4,1,1,1024 4,2,1,1,1024 4,4,1,1,1024; do
  echo "tune $tune"; PDEHIP_TUNE=$tune timeout 200 python tools/time_sizes.py 500x500x300 2>/dev/null | grep "^| 5" | cut -c1-60
done | tee $O/tune_500.log
