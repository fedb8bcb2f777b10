#!/bin/bash
# first GPU session: smoke, parity tests, kernel-variant sweep, bench, rocprof summary
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== microbench"; timeout 300 tools/microbench 512 2>&1 | tee gpurun_out/microbench_512.log | tail -60
echo "== bench"; timeout 600 python bench.py --steps 100 --warmup 10 2>&1 | tail -3 | tee gpurun_out/bench.log
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -3
cd $GRAFT_REPO_ROOT; find gpurun_out/prof1 -name "*stats*" | head; 
for f in $(find gpurun_out/prof1 -name "*kernel_stats.csv" | head -1); do head -12 $f; done
