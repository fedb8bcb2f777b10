#!/bin/bash
# round 6, call 17: kernel trace of the walls benchmark with the coefficient arrays read inside the sweep (PDEHIP_E2_ARRAYS=1) and with the recomputing kernel (=0)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
cd /tmp && export TMPDIR=/tmp
for v in ${ARR_VARIANTS:-1 0}; do
  PDEHIP_E2_ARRAYS=$v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -- python $R/tools/time_bc_program.py 512 100 > /dev/null 2>&1
  echo "== PDEHIP_E2_ARRAYS=$v"; (cd $R; python tools/rocprof_summary.py /tmp/prof_$v gpurun_out/r06_call17_summary_$v.md | cut -c1-230 | head -16)
done | tee $R/gpurun_out/r06_call17_kernel_stats.txt
echo finished
