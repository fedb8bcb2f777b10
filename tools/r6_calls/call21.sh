#!/bin/bash
# round 6, call 21: shell_kernel with the plain-read form for cells without virtual neighbours - parity (faces of time/position, open rows) and its duration
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
python -m pytest tests/test_hip_frows.py tests/test_hip_tails.py tests/test_hip_euler2.py -m gpu -x -q > gpurun_out/r06_call21_tests.log 2>&1; tail -3 gpurun_out/r06_call21_tests.log
python tools/time_bc_program.py 512 200 2>&1 | grep BCPROG | tee gpurun_out/r06_call21_bcprog.log
python tools/time_sizes.py 513 2>/dev/null | tail -4 | tee gpurun_out/r06_call21_sizes.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -- python $R/tools/time_bc_program.py 512 100 > /dev/null 2>&1
(cd $R; python tools/rocprof_summary.py /tmp/prof_s gpurun_out/r06_call21_summary.md | cut -c1-230 | head -14) | tee $R/gpurun_out/r06_call21_kernel_stats.txt
echo finished
