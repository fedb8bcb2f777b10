#!/bin/bash
# round 3, call 28: stress of the world-size-1 slab stepper on tiny grids, with the previous allocation behaviour (no wait after the zero
# fill) and with the current one
O=gpurun_out/r3k
mkdir -p $O
export TMPDIR=/tmp
PDEHIP_LIB=$PWD/tools/libpdehip_diag.so PDEHIP_DIAG_NOWAIT=1 timeout 150 python tools/stress_slab_1d.py 1500 gloo > $O/stress_nowait.log 2>&1
echo "nowait rc=$?"; grep STRESS1D $O/stress_nowait.log | cut -c1-1800 || tail -5 $O/stress_nowait.log
timeout 150 python tools/stress_slab_1d.py 1500 gloo > $O/stress_wait.log 2>&1
echo "wait rc=$?"; grep STRESS1D $O/stress_wait.log | cut -c1-1800 || tail -5 $O/stress_wait.log
