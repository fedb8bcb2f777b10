#!/bin/bash
# round 3, call 9/10: the two-level kernel with overlapped last tiles (odd row lengths / row counts): parity tests + timings
O=gpurun_out/r3i
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_euler2.py tests/test_hip_tails.py tests/test_hip_steppers.py tests/test_hip_distributed.py tests/test_kernel_resources.py tests/test_baseline_configs.py tests/test_expressions.py -m gpu -q --tb=short -p no:cacheprovider --maxfail=30 > $O/pytest.log 2>&1
echo "rc=$?"; grep "passed\|failed" $O/pytest.log | tail -1; grep "^FAILED" $O/pytest.log | head -30
timeout 600 python tools/time_sizes.py 510x510x510 511x511x511 512x512x512 513x513x513 500x500x300 512x513x512 512x512x513 4095x4097 64x64x64 100x100x100 128x128x128 200x200x200 256x256x256 300x300x300 384x384x384 2>&1 | tee $O/time_sizes.log
