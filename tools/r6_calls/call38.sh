#!/bin/bash
# round 6, call 38: rows of three / six chunks in workgroups of three waves (PDEHIP_E2_NWZ3=1) - size table A/B
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
S="384x384x384 512x512x384 512x512x768 256x256x384 640x640x384"
for v in 0 1 0 1; do echo "== PDEHIP_E2_NWZ3=$v"; PDEHIP_E2_NWZ3=$v python tools/time_sizes.py $S 2>/dev/null | grep "float64\|float32" | cut -c1-100; done | tee gpurun_out/r06_call38_nwz3.log
PDEHIP_E2_NWZ3=1 python -m pytest tests/test_hip_euler2.py -m gpu -x -q 2>&1 | tail -2
echo finished
