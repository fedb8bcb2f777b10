#!/bin/bash
# round 3, call 13/15: full GPU suite on the final library of the moment, size table, bench line
O=gpurun_out/r3k
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --maxfail=40 --durations=8 > $O/pytest_gpu.log 2>&1
echo "rc=$?"; grep "passed\|failed" $O/pytest_gpu.log | tail -1; grep "^FAILED" $O/pytest_gpu.log | head -40; grep -A9 "slowest" $O/pytest_gpu.log | cut -c1-150
timeout 600 python tools/time_sizes.py 511x511x511 512x512x512 513x513x513 500x500x300 300x300x300 256x256x256 2>&1 | grep "^|" | tee $O/time_sizes.log
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-400 $O/bench_n1.json
