#!/bin/bash
# round 6, call 24: fp32 diffusion with faces on the wide 4-row tile (euler2_wide4_kernel, PER3 = false) and the stage sweeps of cfg5 on it (PDEHIP_F32_STAGE_WIDE=4): parity, A/B
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
python -m pytest tests/test_hip_euler2.py tests/test_hip_steppers.py tests/test_baseline_configs.py tests/test_hip_frows.py tests/test_hip_tails.py -m gpu -x -q > gpurun_out/r06_call24_tests.log 2>&1; tail -3 gpurun_out/r06_call24_tests.log
for v in 1 0; do echo "== PDEHIP_F32_WIDE4=$v"; PDEHIP_F32_WIDE4=$v python tools/time_f32_walls.py 512 256 500x500x300 2>/dev/null | grep "WALLS.*float32"; done | tee gpurun_out/r06_call24_walls.log
for v in 0 4 1; do echo "== PDEHIP_F32_STAGE_WIDE=$v"; PDEHIP_F32_STAGE_WIDE=$v python bench.py --steps 5 --warmup 2 --no-cpu-baseline --repeats 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d.get("extra",{})
print({k:v for k,v in c.items() if 'cfg5' in k})"; done | tee gpurun_out/r06_call24_cfg5.log
PDEHIP_F32_STAGE_WIDE=4 python -m pytest tests/test_hip_steppers.py tests/test_baseline_configs.py -m gpu -x -q -k "float32 or f32 or cfg5" 2>&1 | tail -2
echo finished
