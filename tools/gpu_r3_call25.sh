#!/bin/bash
# round 3, call 25: the one-level kernel with its last tile moved back (operators, stage sweeps, run-time built kernels on rows that no tile divides)
O=gpurun_out/r3q
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --maxfail=40 > $O/pytest_gpu.log 2>&1
echo "rc=$?"; grep "passed\|failed" $O/pytest_gpu.log | tail -1; grep "^FAILED" $O/pytest_gpu.log | head -40
timeout 600 python tools/time_sizes.py 510x510x510 511x511x511 512x512x512 513x513x513 500x500x300 4095x4097 300x300x300 2>&1 | grep "^|" | tee $O/time_sizes.log
timeout 300 python tools/time_ops.py 2>&1 | grep "513\|511\|512x512x512" | head -20
