#!/bin/bash
# round 3, call 7: stage-operand prefetch A/B (PDEHIP_LIB=nopre = without), block probe with merged face copies
O=gpurun_out/r3g
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_euler2.py tests/test_hip_steppers.py tests/test_hip_distributed.py tests/test_hip_frows.py tests/test_baseline_configs.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest.log 2>&1
echo "rc=$?"; grep "passed\|failed" $O/pytest.log | tail -1; grep "^FAILED" $O/pytest.log | head
for r in 1 2; do
  for v in default nopre; do
    if [ $v = default ]; then unset PDEHIP_LIB; else export PDEHIP_LIB=$R/tools/variants/libpdehip_$v.so; fi
    echo "-- $v"
    timeout 120 python tools/time_ch.py 256 100 float32 2>&1 | grep RKF45
    timeout 120 python tools/time_ch.py 128,256,200 100 float64 2>&1 | grep RKF45
    timeout 120 python tools/time_rk.py 256 cahn_hilliard 2>&1 | grep "rk4_step\|rkf45"
  done
done | tee $O/ab_prefetch.log
unset PDEHIP_LIB
timeout 300 python tools/probe_block.py 256,256,256 300 2>&1 | grep "exchange=" | tee $O/probe_block.log
timeout 300 python tools/probe_block.py 128,128,128 300 2>&1 | grep "exchange=" | tee -a $O/probe_block.log
