#!/bin/bash
# round 5, final build: the whole GPU suite, smoke(), the bench line (driver arguments and defaults), rocprofv3 kernel trace + PMC passes
# (separate runs) of the time loop and of the operator path, the size table, conditions of time and position, the block probe
O=gpurun_out/r5final
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
    d = json.load(open(f"gpurun_out/r5final/{f}.json"))
    print(f, {k: d[k] for k in ("value", "value_best", "ms_per_step")}, "frac", d["roofline"]["frac"], d["roofline"]["frac_best"], "op", d["roofline_operator"]["frac"],
          "nt", d["roofline"].get("nt_copy"), d["roofline"].get("frac_of_nt_copy"), d["roofline_operator"].get("frac_of_nt_copy"), d["roofline"]["copy_ceiling"],
          (d.get("parity") or {}).get("ok"), d.get("extra_error"), d.get("phase_seconds"))
    for k, v in (d.get("roofline_operators") or {}).items():
        print("   op", k, {kk: vv for kk, vv in v.items() if kk in ("kernel_ms", "frac", "nt_copy_gbs", "hipMemcpyDtoD_gbs")} if "kernel_ms" in v or "nt_copy_gbs" in v else {a: (b.get("us_per_step"), b.get("frac")) for a, b in v.items()})
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
echo "== rocprof pmc SQ waits / L2 (bench)"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace -d $R/$O/sq_bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra --repeats 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $R/$O/l2_bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra --repeats 1 > /dev/null 2>&1
cd $R
for t in trace_bench trace_ops; do python tools/rocprof_summary.py $O/$t $O/${t}_summary.md | cut -c1-220 | head -14; done
python tools/rocprof_pmc_summary.py $O/fetch_bench $O/write_bench -o $O/pmc_bench_summary.md | cut -c1-260 | head -12
python tools/rocprof_pmc_summary.py $O/fetch_ops $O/write_ops -o $O/pmc_ops_summary.md | cut -c1-260 | head -8
python tools/rocprof_pmc_summary.py $O/sq_bench -o $O/pmc_sq_summary.md | cut -c1-260 | grep -i "euler2\|kernel |" | head -12
python tools/rocprof_pmc_summary.py $O/l2_bench -o $O/pmc_l2_summary.md | cut -c1-260 | grep -i "euler2\|kernel |" | head -6
find $O -name "*.db" -size +8M -delete
echo "== sizes"
timeout 600 python tools/time_sizes.py 2>/dev/null | tee $O/time_sizes.log | grep "^|" | cut -c1-110
timeout 300 python tools/time_sizes.py 513x513x513 512x512x513 512x512x514 512x512x516 512x512x520 514x514x514 515x515x515 2>/dev/null | tee $O/time_sizes_tails.log | grep "^| 5" | cut -c1-110
echo "== conditions of time and position"
timeout 300 python tools/time_bc_program.py 512 100 2>/dev/null | tee $O/time_bc_program.log | tail -8
echo "== block probe"
timeout 300 python tools/probe_block.py 256,128,512 400 2>&1 | grep "ms/step\|with exchange" | tee $O/probe_block.log
