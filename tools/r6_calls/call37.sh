#!/bin/bash
# round 6, call 37: conditions of time and position on a grid with open rows and columns (order: sweep, open rows with stand-ins, faces with the true coefficients)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
python -m pytest tests/test_hip_frows.py -m gpu -x -q -k "two_steps_per_sweep" 2>&1 | tail -4
echo finished
