#!/bin/bash
# tests + bench + rocprofv3 kernel trace + PMC passes (separate runs, as gpurun requires) + time-skew probe
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_r01.json
echo "== skew probe"; timeout 300 tools/probe_skew 512 2>&1 | tee gpurun_out/probe_skew.log
cd /tmp
echo "== rocprof kernel trace"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_trace -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
echo "== rocprof pmc FETCH_SIZE"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_fetch -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
echo "== rocprof pmc WRITE_SIZE"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_write -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/rocprof_summary.py gpurun_out/prof_trace gpurun_out/prof_trace_summary.md | cut -c1-200 | head -12
python tools/rocprof_pmc_summary.py gpurun_out/prof_fetch gpurun_out/prof_write -o gpurun_out/prof_pmc_summary.md | cut -c1-220
