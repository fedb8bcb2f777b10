#!/bin/bash
# round 6, call 32: is the tall tile the right choice at 300 x 512 x 640 (5 chunks per row: one-wave workgroups)?  default / PDEHIP_EULER2=4, with and without an open row; new parity cases
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
for e in default 4; do echo "== PDEHIP_EULER2=$e"; if [ $e = default ]; then python tools/time_sizes.py 300x512x640 300x513x640 512x512x640 512x513x640 384x513x384 2>/dev/null | grep float64; else PDEHIP_EULER2=$e python tools/time_sizes.py 300x512x640 300x513x640 512x512x640 512x513x640 384x513x384 2>/dev/null | grep float64; fi; done | tee gpurun_out/r06_call32_sizes.log
python -m pytest tests/test_hip_tails.py -m gpu -x -q -k "open_rows_and_open or odd_cubes" 2>&1 | tail -3
echo finished
