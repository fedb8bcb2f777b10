#!/bin/bash
# round 6, call 31: "open" columns of tiles along the rows (open_y: the last 1-4 rows of every plane recomputed by shell_kernel, the tiles cover a
# whole number of them; the tall tile also over open rows) - parity and the size table A/B (PDEHIP_OPEN_Y=0)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
python -m pytest tests/test_hip_tails.py tests/test_hip_euler2.py tests/test_baseline_configs.py tests/test_hip_frows.py -m gpu -x -q > gpurun_out/r06_call31_tests.log 2>&1; tail -3 gpurun_out/r06_call31_tests.log
for v in 1 0; do echo "== PDEHIP_OPEN_Y=$v"; PDEHIP_OPEN_Y=$v python tools/time_sizes.py 513x513x513 514x514x514 515x515x515 516x516x516 512x513x512 512x517x512 300x513x640 2>/dev/null | grep float64; done | tee gpurun_out/r06_call31_sizes.log
echo finished
