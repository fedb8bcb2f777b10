#!/bin/bash
# round 5, call 21: (1) the PF = 2 instances of the one-step kernel, forced on every 3-D test grid, against the oracle; (2) cfg5 with the fp32 stage
# sweeps on the wide tile at one wave per SIMD (PDEHIP_F32_STAGE_WIDE=1) against the narrow tile: parity, then time per attempt
mkdir -p gpurun_out/r5d
cd /root/repo
echo "== PF = 2 forced (PDEHIP_TUNE=2,4,1,2,256)"
PDEHIP_TUNE=2,4,1,2,256 timeout 900 python -m pytest tests/test_hip_operators.py tests/test_hip_derivatives.py tests/test_hip_tails.py tests/test_hip_steppers.py -m gpu -q 2>&1 | grep -E "passed|failed" | tail -2
echo "== defaults (PF = 2 for large fp64 fields)"
timeout 900 python -m pytest tests/test_baseline_configs.py tests/test_hip_operators.py tests/test_hip_properties.py -m gpu -q 2>&1 | grep -E "passed|failed" | tail -2
echo "== fp32 stage sweeps on the wide one-wave tile: parity"
PDEHIP_F32_STAGE_WIDE=1 timeout 900 python -m pytest tests/test_baseline_configs.py tests/test_hip_steppers.py tests/test_hip_euler2.py tests/test_hip_adaptive_euler.py tests/test_hip_properties.py -m gpu -q 2>&1 | grep -E "passed|failed" | tail -2
L=gpurun_out/r5d/cfg5_stage_wide.log
: > $L
for rep in 1 2 3; do
for w in 0 1; do
  echo -n "PDEHIP_F32_STAGE_WIDE=$w : " >> $L
  PDEHIP_F32_STAGE_WIDE=$w timeout 300 python tools/time_cfg5.py 2>/dev/null | grep CFG5 >> $L
done
done
cat $L
python tools/time_lap.py 512 2>/dev/null | grep LAP
