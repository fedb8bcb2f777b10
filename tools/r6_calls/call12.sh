#!/bin/bash
# round 6, call 12: the transposed tail-columns kernel for open rows on all-periodic grids - parity, then the size table with / without it
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_euler2.py tests/test_hip_tails.py tests/test_hip_properties.py -x -q > gpurun_out/r06_call12_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/r06_call12_tests.log | tail -3
{
echo "## with the tail-columns kernel (default)"
timeout 900 python tools/time_sizes.py 512x512x512 513x513x513 515x515x515 512x512x513 512x512x520 514x514x514 2>/dev/null
echo "## PDEHIP_TAIL_COLUMNS=0 (the LDS-tiled shell kernel of rounds 4-5)"
PDEHIP_TAIL_COLUMNS=0 timeout 900 python tools/time_sizes.py 513x513x513 515x515x515 512x512x513 512x512x520 2>/dev/null
} | tee gpurun_out/r06_call12_time_sizes.md
