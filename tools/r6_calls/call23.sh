#!/bin/bash
# round 6, call 23: fp32 all-periodic diffusion on the wide 4-row tile at one wave per SIMD (euler2_wide4_per_kernel) - parity and A/B
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
python -m pytest tests/test_hip_euler2.py -m gpu -x -q > gpurun_out/r06_call23_tests.log 2>&1; tail -3 gpurun_out/r06_call23_tests.log
for v in 1 0; do echo "== PDEHIP_F32_WIDE4=$v"; PDEHIP_F32_WIDE4=$v python tools/time_sizes.py 512x512x512 384x384x384 192x192x256 64x64x256 128x128x256 1024x256x256 256x256x512 640x640x512 100x100x256 2>/dev/null | grep float32; done | tee gpurun_out/r06_call23_sizes.log
echo finished
