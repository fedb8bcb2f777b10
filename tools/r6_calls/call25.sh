#!/bin/bash
# round 6, call 25: wide 4-row fp32 tile with plain stores at 512^3 (PDEHIP_F32_WIDE4_NT=0), faces and all-periodic; cfg5 with the stage sweeps on the wide 4-row tile
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
for v in 1 0; do echo "== PDEHIP_F32_WIDE4_NT=$v"; PDEHIP_F32_WIDE4_NT=$v python tools/time_f32_walls.py 512 640x512x512 2>/dev/null | grep "WALLS.*float32"; PDEHIP_F32_WIDE4_NT=$v python tools/time_sizes.py 512x512x512 640x640x512 2>/dev/null | grep float32; done | tee gpurun_out/r06_call25_nt.log
for v in 0 4; do echo "== PDEHIP_F32_STAGE_WIDE=$v"; PDEHIP_F32_STAGE_WIDE=$v python bench.py --steps 5 --warmup 2 --no-cpu-baseline --repeats 1 2>/dev/null | python tools/print_extra.py cfg5; done | tee gpurun_out/r06_call25_cfg5.log
echo finished
