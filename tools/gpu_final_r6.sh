#!/bin/bash
# round 6, final build: the whole GPU suite, smoke(), rocprofv3 kernel trace + PMC passes (separate runs) of the time loop and of the operator
# path, profiles/traffic.json refreshed from those passes (keyed by the kernel instance the library reports), then the bench lines (driver
# arguments and defaults; they pick the new traffic entry up), the size table, the slab probes, the fastmath A/B
O=gpurun_out/r6final
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
echo "== pytest -m gpu (everything)"
timeout 3000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_all.log 2>&1
echo "rc=$?"; grep -E "passed|failed" $O/gpu_all.log | tail -2; grep "^FAILED\|^ERROR" $O/gpu_all.log | head
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -i "smoke" | tail -2
cd /tmp
echo "== rocprof kernel trace (bench: the time loop and the dominant kernel only - every launch of the kernel is a 512^3 launch; then the whole line)"
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/trace_bench -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra --repeats 3 > $R/$O/trace_bench.json 2>/dev/null
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/trace_bench_full -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --repeats 3 > $R/$O/trace_bench_full.json 2>/dev/null
echo "== rocprof pmc FETCH_SIZE / WRITE_SIZE (bench)"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/fetch_bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra --repeats 1 > $R/$O/fetch_bench.json 2>/dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/write_bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra --repeats 1 > /dev/null 2>&1
echo "== rocprof kernel trace + pmc (operators)"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace_ops -- python $R/tools/run_laplace.py > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/fetch_ops -- python $R/tools/run_laplace.py > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/write_ops -- python $R/tools/run_laplace.py > /dev/null 2>&1
echo "== rocprof pmc SQ waits / instructions (bench; then with the round-5 instance: PDEHIP_E2_PER3=0)"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace -d $R/$O/sq_bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra --repeats 1 > /dev/null 2>&1
PDEHIP_E2_PER3=0 PDEHIP_EULER2=8 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace -d $R/$O/sq_bench_r5tile -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra --repeats 1 > /dev/null 2>&1
PDEHIP_FASTMATH=1 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace -d $R/$O/sq_bench_fastmath -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra --repeats 1 > /dev/null 2>&1
cd $R
for t in trace_bench trace_bench_full trace_ops; do python tools/rocprof_summary.py $O/$t $O/${t}_summary.md | cut -c1-220 | head -14; done
python tools/rocprof_pmc_summary.py $O/fetch_bench $O/write_bench -o $O/pmc_bench_summary.md | cut -c1-260 | head -10
python tools/rocprof_pmc_summary.py $O/fetch_ops $O/write_ops -o $O/pmc_ops_summary.md | cut -c1-260 | head -8
for v in sq_bench sq_bench_r5tile sq_bench_fastmath; do python tools/rocprof_pmc_summary.py $O/$v -o $O/pmc_${v}_summary.md | cut -c1-260 | grep -i "euler2" | head -6; done
python tools/update_traffic.py $O/fetch_bench $O/write_bench $O/fetch_bench.json "profiles/r06_final_rocprof_pmc.md (final round-6 build, tools/gpu_final_r6.sh; another launch than a given bench run)" | tee $O/traffic_update.log
cp profiles/traffic.json $O/traffic.json
find $O -name "*.db" -size +8M -delete
echo "== bench (driver arguments, then default, then fastmath)"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench20.err | tail -1 > $O/bench20.json
timeout 1200 python bench.py 2> $O/bench.err | tail -1 > $O/bench_n1.json
PDEHIP_FASTMATH=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs 2> /dev/null | tail -1 > $O/bench20_fastmath.json
python - <<'PY'
import json
for f in ("bench20", "bench_n1", "bench20_fastmath"):
    d = json.load(open(f"gpurun_out/r6final/{f}.json"))
    print(f, {k: d[k] for k in ("value", "value_best", "ms_per_step")}, "frac", d["roofline"]["frac"], d["roofline"]["frac_best"], d["roofline"]["kernel"][:90], "traffic", d["roofline"]["traffic"],
          "op", d["roofline_operator"]["frac"], "nt", d["roofline"].get("nt_copy"), d["roofline"].get("frac_of_nt_copy"), (d.get("parity") or {}).get("ok"), d.get("extra_error"), d.get("phase_seconds"))
    for k, v in (d.get("extra") or {}).items():
        print("   ", k, json.dumps(v)[:400])
PY
echo "== sizes"
timeout 900 python tools/time_sizes.py 2>/dev/null | tee $O/time_sizes.log | grep "^|" | cut -c1-110
timeout 600 python tools/time_sizes.py 513x513x513 512x512x513 512x512x514 512x512x520 514x514x514 515x515x515 2>/dev/null | tee $O/time_sizes_tails.log | grep "^| 5" | cut -c1-110
echo "== slab probe"
for s in 64,512,512 128,512,512 256,512,512; do timeout 300 python tools/probe_slab.py $s 400 2>&1 | grep "slab stepper\|euler_run"; done | tee $O/probe_slab.log
echo "== block probe"
timeout 300 python tools/probe_block.py 256,128,512 400 2>&1 | grep "ms/step\|with exchange" | tee $O/probe_block.log
