#!/bin/bash
# A/B timing of two builds of libpdehip.so on the SAME box (boxes differ by +-5 %, so only same-session numbers compare).
# usage: bash tools/ab_time.sh <path-to-other-libpdehip.so>      (A = the in-tree build, B = the other one, via PDEHIP_LIB)
B=${1:?path to the other libpdehip.so}
for r in 1 2 3; do
  timeout 100 python tools/time_euler2.py 512 200 2>&1 | tail -1 | sed 's/^/A /'
  PDEHIP_ALLOW_LIB_OVERRIDE=1 PDEHIP_LIB=$B timeout 100 python tools/time_euler2.py 512 200 2>&1 | tail -1 | sed 's/^/B /'
done
for n in 256 128,512,512; do
  timeout 100 python tools/time_euler2.py $n 300 2>&1 | tail -1 | sed 's/^/A /'
  PDEHIP_ALLOW_LIB_OVERRIDE=1 PDEHIP_LIB=$B timeout 100 python tools/time_euler2.py $n 300 2>&1 | tail -1 | sed 's/^/B /'
done
