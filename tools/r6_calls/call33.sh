#!/bin/bash
# round 6, call 33: the tall tile only where the chunks of a row come in fours - size table with the new rule, and forced tall (PDEHIP_EULER2=8) / forced 4-row (=4) for comparison
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/py-pde_amd
S="512x512x512 512x512x256 512x512x768 512x512x1024 300x512x640 384x384x384 384x513x384 513x513x513 640x640x512"
for e in default 8 4; do echo "== PDEHIP_EULER2=$e"; if [ $e = default ]; then python tools/time_sizes.py $S 2>/dev/null | grep float64; else PDEHIP_EULER2=$e python tools/time_sizes.py $S 2>/dev/null | grep float64; fi; done | tee gpurun_out/r06_call33_sizes.log
echo finished
