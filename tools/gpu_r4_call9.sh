#!/bin/bash
# round 4, call 9: where the time of the split rows goes (kernel trace at 513^3), tails tests again
O=gpurun_out/r4i
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests/test_hip_tails.py tests/test_hip_operators.py -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_tails.log 2>&1
echo "rc=$?"; tail -2 $O/gpu_tails.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace513 -- python $R/tools/time_sizes.py 513x513x513 512x512x512 > $R/$O/t513.log 2>/dev/null
cd $R
python tools/rocprof_summary.py $O/trace513 $O/trace513_summary.md | cut -c1-200 | head -14
grep "^|" $O/t513.log
find $O -name "*.db" -size +8M -delete
