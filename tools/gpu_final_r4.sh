#!/bin/bash
# round 4, final build: the whole GPU suite, smoke(), the bench line (driver arguments and defaults), rocprofv3 kernel trace + PMC passes
# (separate runs) of the time loop and of the operator path, the size table
O=gpurun_out/r4final
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
echo "== pytest -m gpu (everything)"
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_all.log 2>&1
echo "rc=$?"; tail -3 $O/gpu_all.log; grep "^FAILED\|^ERROR" $O/gpu_all.log | head
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench (driver arguments, then default)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench20.err | tail -1 > $O/bench20.json
timeout 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench_n1.json
python - <<'PY'
import json
for f in ("bench20", "bench_n1"):
    d = json.load(open(f"gpurun_out/r4final/{f}.json"))
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
echo "== sizes"
timeout 600 python tools/time_sizes.py 2>/dev/null | tee $O/time_sizes.log | grep "^|" | cut -c1-110
timeout 300 python tools/time_sizes.py 513x513x513 512x512x513 512x512x514 512x512x516 512x512x520 514x514x514 515x515x515 2>/dev/null | tee $O/time_sizes_tails.log | grep "^| 5" | cut -c1-110
