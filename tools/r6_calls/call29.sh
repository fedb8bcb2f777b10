#!/bin/bash
# round 6, call 29: kernel trace of tools/time_f32_walls.py (fp64 part) with the tall tile for faces and without
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
cd /tmp && export TMPDIR=/tmp
for v in 1 0 1 0; do
  rm -rf /tmp/prof_t
  PDEHIP_E2_TALL_FACES=$v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -- python $R/tools/time_f32_walls.py 512 > /tmp/walls_$v.log 2>/dev/null
  echo "== PDEHIP_E2_TALL_FACES=$v"; grep WALLS /tmp/walls_$v.log; (cd $R; python tools/rocprof_summary.py /tmp/prof_t gpurun_out/r06_call29_summary_$v.md | grep -E "euler2" | cut -c1-230)
done | tee $R/gpurun_out/r06_call29_kernel_stats.txt
echo finished
