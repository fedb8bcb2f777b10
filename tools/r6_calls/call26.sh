#!/bin/bash
# round 6, call 26: the tall fp64 tile WITH faces in its three-buffer forms (PDEHIP_EULER2=8 + PDEHIP_E2_TALL_TW=2/3) against the 4-row tile (default) - walls, 512^3 / 384^3 / 256^3
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
for cfg in "default" "8,0" "8,2" "8,3"; do
  echo "== $cfg"
  if [ "$cfg" = "default" ]; then python tools/time_f32_walls.py 512 384 256 2>/dev/null | grep "WALLS.*float64";
  else PDEHIP_EULER2=8 PDEHIP_E2_TALL_TW=${cfg#8,} python tools/time_f32_walls.py 512 384 256 2>/dev/null | grep "WALLS.*float64"; fi
done | tee gpurun_out/r06_call26_tall_faces.log
PDEHIP_EULER2=8 PDEHIP_E2_TALL_TW=3 python -m pytest tests/test_hip_euler2.py -m gpu -x -q 2>&1 | tail -2
PDEHIP_EULER2=8 PDEHIP_E2_TALL_TW=2 python -m pytest tests/test_hip_euler2.py -m gpu -x -q 2>&1 | tail -2
echo finished
