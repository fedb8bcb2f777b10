#!/bin/bash
# round 4, call 20: shell kernel with index-based operand access - parity, duration, instruction count
O=gpurun_out/r4t
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_hip_frows.py tests/test_hip_euler2.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/tests.log 2>&1
echo "tests rc=$?"; tail -2 $O/tests.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace -- python $R/tools/run_bc_program.py 512 40 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace -d $R/$O/valu -- python $R/tools/run_bc_program.py 512 20 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU --kernel-trace -d $R/$O/busy -- python $R/tools/run_bc_program.py 512 20 > /dev/null 2>&1
cd $R
python tools/rocprof_summary.py $O/trace $O/trace_summary.md | grep -i "shell\|euler2\|bc_refresh\|kernel |" | cut -c1-200
python tools/rocprof_pmc_summary.py $O/valu $O/busy | grep -i "shell\|kernel |" | cut -c1-260
find $O -name "*.db" -size +8M -delete
for k in 1 2; do timeout 600 python tools/time_bc_program.py 512 100 2>/dev/null | grep BCPROG | cut -c1-260; done | tee $O/bcprog.log
