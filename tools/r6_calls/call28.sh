#!/bin/bash
# round 6, call 28: kernel trace of the walls benchmark with the tall tile for faces (PDEHIP_E2_TALL_FACES=1, default) and without (=0)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
cd /tmp && export TMPDIR=/tmp
for v in 1 0 1 0; do
  rm -rf /tmp/prof_t
  PDEHIP_E2_TALL_FACES=$v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -- python $R/tools/time_bc_program.py 512 100 > /dev/null 2>&1
  echo "== PDEHIP_E2_TALL_FACES=$v"; (cd $R; python tools/rocprof_summary.py /tmp/prof_t gpurun_out/r06_call28_summary_$v.md | grep -E "euler2|shell" | cut -c1-230)
done | tee $R/gpurun_out/r06_call28_kernel_stats.txt
echo finished
