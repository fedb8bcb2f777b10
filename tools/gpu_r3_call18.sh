#!/bin/bash
# round 3, call 18: deferred fix-up of loaded planes (-DPDEHIP_E2_DEFER=1) A/B on the Runge-Kutta sweeps and the Euler sweeps
O=gpurun_out/r3n
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
for r in 1 2; do
  for v in default defer; do
    if [ $v = default ]; then unset PDEHIP_LIB; else export PDEHIP_LIB=$R/tools/variants/libpdehip_$v.so; fi
    echo "-- $v"
    timeout 120 python tools/time_ch.py 256 100 float32 2>&1 | grep "RKF45\|Euler"
    timeout 120 python tools/time_ch.py 128,256,200 100 float64 2>&1 | grep "RKF45\|Euler"
    timeout 120 python tools/time_ch.py 256 100 float64 2>&1 | grep "RKF45\|Euler"
    timeout 200 python tools/time_sizes.py 512x512x512 513x513x513 256x256x256 2>&1 | grep "float"
  done
done | tee $O/ab_defer.log
unset PDEHIP_LIB
