#!/bin/bash
# round 6, call 30: kernel trace of fp32 walls (tools/time_f32_walls.py) with the wide 4-row tile and with the 2-row tile, 512^3 and 256^3
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
cd /tmp && export TMPDIR=/tmp
for n in 512 256; do for v in 1 0 1 0; do
  rm -rf /tmp/prof_t
  PDEHIP_F32_WIDE4=$v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -- python $R/tools/time_f32_walls.py $n > /dev/null 2>&1
  echo "== n=$n PDEHIP_F32_WIDE4=$v"; (cd $R; python tools/rocprof_summary.py /tmp/prof_t gpurun_out/r06_call30_summary.md | grep -E "euler2.*float" | cut -c1-230)
done; done | tee $R/gpurun_out/r06_call30_kernel_stats.txt
echo finished
