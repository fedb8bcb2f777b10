#!/bin/bash
# round 6, call 40: fp32 on the fast block loop with a cut fastest axis (interior box on the narrow tile) - to-self parity, block probe fp32
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
python -m pytest tests/test_hip_distributed.py tests/test_hip_euler2.py -m gpu -x -q > gpurun_out/r06_call40_tests.log 2>&1; grep -E "passed|failed|^FAILED|Error" gpurun_out/r06_call40_tests.log | tail -5
echo finished
