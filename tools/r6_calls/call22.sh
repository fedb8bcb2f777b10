#!/bin/bash
# round 6, call 22: shell_kernel variants (PDEHIP_SHELL_X: bit 0 plain-read form, bit 1 face axis slowest in the thread maps) - duration in the kernel trace
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3; do
  PDEHIP_SHELL_X=$v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_s$v -- python $R/tools/time_bc_program.py 512 100 > /dev/null 2>&1
  echo "== PDEHIP_SHELL_X=$v"; (cd $R; python tools/rocprof_summary.py /tmp/prof_s$v gpurun_out/r06_call22_summary_$v.md | grep shell_kernel | cut -c1-230)
done | tee $R/gpurun_out/r06_call22_kernel_stats.txt
echo finished
