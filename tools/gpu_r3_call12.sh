#!/bin/bash
# round 3, call 12: conditions that read the field (ABI 3) on the real library, real py-pde subset, kernel timeline of the slab loop
O=gpurun_out/r3j
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_frows.py tests/test_hip_steppers.py tests/test_hip_distributed.py tests/test_expressions.py tests/test_hip_bcs.py -m gpu -q --tb=short -p no:cacheprovider --maxfail=20 > $O/pytest.log 2>&1
echo "rc=$?"; grep "passed\|failed" $O/pytest.log | tail -1; grep "^FAILED" $O/pytest.log | head -20
if [ -d _refscratch ]; then
  PDEHIP_REFERENCE=$R/_refscratch PDEHIP_DROPIN_REAL=1 timeout 900 python -m pytest tests/test_pypde_dropin.py -q --tb=short -p no:cacheprovider -k "nonlinearly or time_dependent or user_funcs or consistency" 2>&1 | tail -5
fi
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/$O/trace_slab -- python $R/tools/probe_slab.py 64,512,512 100 > $R/$O/probe_slab.log 2>&1
cd $R
grep "ms/step" $O/probe_slab.log
TIMELINE_SKIP='fillBuffer|copyBuffer|layout_copy|ghost_kernel' python tools/rocprof_timeline.py $O/trace_slab 150 48 | cut -c1-150 | tee $O/timeline_slab.txt
find $O -name "*.db" -size +8M -delete
