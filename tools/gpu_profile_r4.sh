#!/bin/bash
# round 4: bench line + rocprofv3 kernel trace + PMC passes (separate runs, as gpurun requires) for the time loop (euler2_kernel) and for the
# operator path (lap_march_kernel); op timings with the corrected ghost-cell line; new GPU tests of this round
O=gpurun_out/r4prof
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
echo "== new / changed GPU tests"
timeout 900 python -m pytest tests/test_hip_frows.py tests/test_hip_complex.py tests/test_hip_operators.py -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_new.log 2>&1
echo "rc=$?"; tail -3 $O/gpu_new.log; grep "^FAILED\|^ERROR" $O/gpu_new.log | head
echo "== bench (driver arguments, then default)"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench20.err | tail -1 > $O/bench20.json
timeout 600 python bench.py 2> $O/bench.err | tail -1 > $O/bench_n1.json
python - <<'PY'
import json
for f in ("bench20", "bench_n1"):
    d = json.load(open(f"gpurun_out/r4prof/{f}.json"))
    print(f, {k: d[k] for k in ("value", "value_best", "ms_per_step")}, "frac", d["roofline"]["frac"], d["roofline"]["frac_best"], "op", d["roofline_operator"]["frac"],
          d["roofline"]["copy_ceiling"], d.get("parity", {}).get("ok"), d.get("extra_error"))
    for k, v in (d.get("extra") or {}).items():
        print("   ", k, v)
PY
cd /tmp
echo "== rocprof kernel trace (bench)"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace_bench -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --repeats 3 > $R/$O/trace_bench.json 2>/dev/null
echo "== rocprof pmc FETCH_SIZE / WRITE_SIZE (bench)"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/fetch_bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra --repeats 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/write_bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra --repeats 1 > /dev/null 2>&1
echo "== rocprof kernel trace + pmc (operators)"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace_ops -- python $R/tools/run_laplace.py > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/fetch_ops -- python $R/tools/run_laplace.py > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/write_ops -- python $R/tools/run_laplace.py > /dev/null 2>&1
cd $R
for t in trace_bench trace_ops; do python tools/rocprof_summary.py $O/$t $O/${t}_summary.md | cut -c1-220 | head -12; done
python tools/rocprof_pmc_summary.py $O/fetch_bench $O/write_bench -o $O/pmc_bench_summary.md | cut -c1-260 | head -12
python tools/rocprof_pmc_summary.py $O/fetch_ops $O/write_ops -o $O/pmc_ops_summary.md | cut -c1-260 | head -8
find $O -name "*.db" -size +8M -delete
echo "== op timings (ghost-cell line corrected)"
timeout 300 python tools/time_ops.py 2>&1 | grep -i "ghost\|grid\|---" | tee $O/time_ops_ghost.log
