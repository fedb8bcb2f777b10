#!/bin/bash
# round 3: bench line + rocprofv3 kernel trace + PMC passes (separate runs, as gpurun requires) for the time loop (euler2_kernel)
# and for the operator path (lap_march_kernel)
O=gpurun_out/r3prof
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
echo "== bench"; timeout 600 python bench.py 2> $O/bench.err | tail -1 | tee $O/bench_n1.json | cut -c1-400
cd /tmp
echo "== rocprof kernel trace (bench)"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace_bench -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/$O/trace_bench.json 2>/dev/null
echo "== rocprof pmc FETCH_SIZE / WRITE_SIZE (bench)"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/fetch_bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/write_bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
echo "== rocprof kernel trace + pmc (operators)"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace_ops -- python $R/tools/run_laplace.py > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/fetch_ops -- python $R/tools/run_laplace.py > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/write_ops -- python $R/tools/run_laplace.py > /dev/null 2>&1
cd $R
for t in trace_bench trace_ops; do python tools/rocprof_summary.py $O/$t $O/${t}_summary.md | cut -c1-220 | head -12; done
python tools/rocprof_pmc_summary.py $O/fetch_bench $O/write_bench -o $O/pmc_bench_summary.md | cut -c1-260
python tools/rocprof_pmc_summary.py $O/fetch_ops $O/write_ops -o $O/pmc_ops_summary.md | cut -c1-260
# keep the merge small: the raw databases are not needed back
find $O -name "*.db" -size +8M -delete
echo "== cfg5 kernel trace (256^3 fp32 RKF45 through eq.solve)"
cd /tmp
ONLY=cfg5 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace_cfg5 -- python $R/tools/bench_configs.py > $R/$O/cfg5_bench.log 2>/dev/null
cd $R
python tools/rocprof_summary.py $O/trace_cfg5 $O/trace_cfg5_summary.md | cut -c1-220 | head -12
tail -2 $O/cfg5_bench.log
echo "== cfg5 long parity numbers"
timeout 900 python -m pytest tests/test_baseline_configs.py -m gpu -q -s -k "cfg5 and long or 10k" -p no:cacheprovider 2>&1 | grep "cfg5\|cfg3\|passed\|failed" | tee $O/cfg5_long.log
find $O -name "*.db" -size +8M -delete
