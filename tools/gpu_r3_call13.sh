#!/bin/bash
# round 3, call 13: full GPU suite on the library with ABI 3 (conditions that read the field), bench line
O=gpurun_out/r3k
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --maxfail=40 > $O/pytest_gpu.log 2>&1
echo "rc=$?"; grep "passed\|failed" $O/pytest_gpu.log | tail -1; grep "^FAILED" $O/pytest_gpu.log | head -40
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-1500 $O/bench_n1.json
