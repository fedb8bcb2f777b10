#!/bin/bash
# round 4, call 11: shell kernel with static axes - kernel trace of the time-dependent run; strip with the columns in neighbouring lanes
O=gpurun_out/r4k
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_hip_tails.py tests/test_hip_frows.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/gpu_tests.log 2>&1
echo "rc=$?"; tail -3 $O/gpu_tests.log
echo "== bc program"; timeout 600 python tools/time_bc_program.py 512 100 2>/dev/null | tee $O/bcprog_two.log | grep BCPROG
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_bc -- python $R/tools/time_bc_program.py 512 20 > $R/$O/trace_bc.log 2>/dev/null
cd $R
python tools/rocprof_summary.py $O/trace_bc $O/trace_bc_summary.md | cut -c1-220 | head -14
echo skip-sizes
find $O -name "*.db" -size +8M -delete
