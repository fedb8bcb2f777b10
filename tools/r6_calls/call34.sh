#!/bin/bash
# round 6, call 34: open tile columns for the fp32 wide 4-row tile; parity (tails, euler2, baseline configs, frows) and the size table A/B (PDEHIP_OPEN_Y=0)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
python -m pytest tests/test_hip_tails.py tests/test_hip_euler2.py tests/test_baseline_configs.py tests/test_hip_frows.py -m gpu -x -q > gpurun_out/r06_call34_tests.log 2>&1; tail -3 gpurun_out/r06_call34_tests.log
for v in 1 0; do echo "== PDEHIP_OPEN_Y=$v"; PDEHIP_OPEN_Y=$v python tools/time_sizes.py 517x517x517 519x519x519 512x517x512 513x513x513 514x514x514 515x515x515 300x513x640 2>/dev/null | grep float; done | tee gpurun_out/r06_call34_sizes.log
echo finished
