#!/bin/bash
# round 6, call 16: faces given as coefficient arrays inside the two-step sweep (euler2_arr_kernel) - parity + the walls benchmark
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/py-pde_amd
python -m pytest tests/test_hip_frows.py -m gpu -x -q > gpurun_out/r06_call16_tests.log 2>&1; tail -5 gpurun_out/r06_call16_tests.log
for v in 1 0; do echo "== PDEHIP_E2_ARRAYS=$v"; PDEHIP_E2_ARRAYS=$v python tools/time_bc_program.py 512 200; done 2>&1 | grep -E "==|BCPROG" | tee gpurun_out/r06_call16_bcprog.log
echo finished
