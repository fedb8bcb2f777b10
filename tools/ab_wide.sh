#!/bin/bash
# A/B of two builds over the other users of the two-level kernel (same box): Cahn-Hilliard fp64 / fp32, 2-D, RK steps.
# usage: bash tools/ab_wide.sh <path-to-other-libpdehip.so>     (A = in-tree build, B = the other one)
B=${1:?path to the other libpdehip.so}
run() { "$@" 2>&1 | grep -v "amdgpu.ids" | tail -${N:-1} | sed "s/^/$TAG /"; }
for cfg in "512 50 float64" "256 200 float32" "512,512 2000 float64"; do
  TAG=A N=2 run timeout 200 python tools/time_ch.py $cfg
  TAG=B N=2 PDEHIP_ALLOW_LIB_OVERRIDE=1 PDEHIP_LIB=$B run timeout 200 python tools/time_ch.py $cfg
done
for n in 4096,4096 1024,1024; do
  TAG=A run timeout 100 python tools/time_euler2.py $n 500
  TAG=B PDEHIP_ALLOW_LIB_OVERRIDE=1 PDEHIP_LIB=$B run timeout 100 python tools/time_euler2.py $n 500
done
TAG=A N=3 run timeout 200 python tools/time_rk.py 512 cahn_hilliard periodic
TAG=B N=3 PDEHIP_ALLOW_LIB_OVERRIDE=1 PDEHIP_LIB=$B run timeout 200 python tools/time_rk.py 512 cahn_hilliard periodic
