#!/bin/bash
# round 3, call 17: Runge-Kutta stage operands through LDS (direct-to-LDS loads one plane ahead): parity + A/B against -DPDEHIP_STAGE_LDS=0
O=gpurun_out/r3m
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_steppers.py tests/test_hip_euler2.py tests/test_hip_tails.py tests/test_baseline_configs.py tests/test_hip_properties.py tests/test_hip_distributed.py tests/test_hip_frows.py -m gpu -q --tb=short -p no:cacheprovider --maxfail=30 > $O/pytest.log 2>&1
echo "rc=$?"; grep "passed\|failed" $O/pytest.log | tail -1; grep "^FAILED" $O/pytest.log | head -30
for r in 1 2; do
  for v in default nolds; do
    if [ $v = default ]; then unset PDEHIP_LIB; else export PDEHIP_LIB=$R/tools/variants/libpdehip_$v.so; fi
    echo "-- $v"
    timeout 120 python tools/time_ch.py 256 100 float32 2>&1 | grep "RKF45\|Euler"
    timeout 120 python tools/time_ch.py 128,256,200 100 float64 2>&1 | grep RKF45
    timeout 120 python tools/time_ch.py 256 100 float64 2>&1 | grep RKF45
    timeout 120 python tools/time_rk.py 256 cahn_hilliard 2>&1 | grep "rk4_step\|rkf45"
  done
done | tee $O/ab_stage_lds.log
unset PDEHIP_LIB
