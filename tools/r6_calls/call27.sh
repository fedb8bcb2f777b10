#!/bin/bash
# round 6, call 27: the tall tile with faces by default for fields beyond 400 MB - parity test, walls timings (fp64, time/position-dependent faces too)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
python -m pytest tests/test_baseline_configs.py tests/test_hip_frows.py tests/test_hip_euler2.py -m gpu -x -q > gpurun_out/r06_call27_tests.log 2>&1; tail -3 gpurun_out/r06_call27_tests.log
for v in 1 0; do echo "== PDEHIP_E2_TALL_FACES=$v"; PDEHIP_E2_TALL_FACES=$v python tools/time_f32_walls.py 512 640x512x512 2>/dev/null | grep "WALLS.*float64"; PDEHIP_E2_TALL_FACES=$v python tools/time_bc_program.py 512 200 2>&1 | grep BCPROG; done | tee gpurun_out/r06_call27_walls.log
echo finished
