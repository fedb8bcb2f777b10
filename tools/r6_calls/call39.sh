#!/bin/bash
# round 6, call 39: the slab sweeps of all-periodic grids on the all-periodic 4-row body (euler2_peryz_kernel) - distributed parity to self, slab probes A/B (PDEHIP_E2_PERYZ=0)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
python -m pytest tests/test_hip_distributed.py tests/test_hip_euler2.py -m gpu -x -q > gpurun_out/r06_call39_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r06_call39_tests.log | tail -2
for v in 1 0 1 0; do echo "== PDEHIP_E2_PERYZ=$v"; for s in 64,512,512 128,512,512 256,512,512; do PDEHIP_E2_PERYZ=$v timeout 300 python tools/probe_slab.py $s 400 2>&1 | grep "exchange=True"; done; done | tee gpurun_out/r06_call39_probe.log
echo finished
