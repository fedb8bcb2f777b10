#!/bin/bash
# bench + rocprofv3 kernel trace + PMC passes (separate runs, as gpurun requires)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_r01.json
cd /tmp
echo "== rocprof kernel trace"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_trace -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
echo "== rocprof pmc FETCH_SIZE"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_fetch -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
echo "== rocprof pmc WRITE_SIZE"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_write -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/rocprof_summary.py gpurun_out/prof_trace gpurun_out/prof_trace_summary.md | cut -c1-200 | head -12
python tools/rocprof_pmc_summary.py gpurun_out/prof_fetch gpurun_out/prof_write -o gpurun_out/prof_pmc_summary.md | cut -c1-220
