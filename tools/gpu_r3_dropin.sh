#!/bin/bash
# round 3, call 1 (run through gpurun after tools/ship_reference.sh): the py-pde plugin class against the REAL libpdehip.so on the
# MI355X, the reference timed on this box's host cores, the new f-row GPU tests, copy-ceiling calibration, bench line, cfg5 profile.
O=gpurun_out/r3a
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
export PDEHIP_REFERENCE=$R/_refscratch PDEHIP_DROPIN_REAL=1 PDEHIP_DROPIN_LOG=$R/$O/dropin_outcomes.txt
rm -f $PDEHIP_DROPIN_LOG
echo "== drop-in tests, real library"
timeout 1500 python -m pytest tests/test_pypde_dropin.py tests/test_pypde_plugin.py tests/test_class_pde_fuzz.py tests/test_expression_fuzz.py \
    tests/test_reference_suite.py -q -rA --tb=short -p no:cacheprovider > $O/dropin_pytest.log 2>&1
echo "rc=$?"; tail -3 $O/dropin_pytest.log
grep -c "^PASSED" $O/dropin_pytest.log; grep "^FAILED\|^ERROR" $O/dropin_pytest.log | head -40
grep -c "^PASSED" $PDEHIP_DROPIN_LOG; grep "^LOADED" $PDEHIP_DROPIN_LOG | sort | uniq -c
echo "== slab solver of the plugin at world size 1 (RCCL to self)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29611 tests/pypde_slab_worker.py > $O/slab_worker.log 2>&1
echo "rc=$?"; grep PYPDESLAB $O/slab_worker.log | cut -c1-600; tail -3 $O/slab_worker.log | cut -c1-300
echo "== reference on this box's host cores"
PDEHIP_WHERE="MI355X box host cores (reference shipped as untracked scratch for this one run)" timeout 900 python tools/time_reference_cpu.py 512 6 $R/$O/reference_cpu_gpubox.json 2>&1 | tail -4
unset PDEHIP_DROPIN_REAL PDEHIP_REFERENCE PDEHIP_DROPIN_LOG
echo "== new gpu tests"
timeout 1200 python -m pytest tests/test_hip_frows.py tests/test_baseline_configs.py -m gpu -q -rA --tb=short -p no:cacheprovider > $O/frows_pytest.log 2>&1
echo "rc=$?"; tail -3 $O/frows_pytest.log; grep "^FAILED\|^ERROR" $O/frows_pytest.log | head
echo "== copy ceiling"
timeout 300 tools/microbench4 > $O/microbench4.log 2>&1; tail -45 $O/microbench4.log
echo "== bench (with clock / power samples)"
(while true; do rocm-smi --showclocks --showpower --json 2>/dev/null | head -c 2000; echo; sleep 0.5; done) > $O/smi_during_bench.log &
SMI=$!
timeout 600 python bench.py 2> $O/bench.err | tail -1 > $O/bench_n1.json
kill $SMI
cut -c1-1500 $O/bench_n1.json
echo "== cfg5 kernel trace"
cd /tmp
ONLY=cfg5 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace_cfg5 -- python $R/tools/bench_configs.py > $R/$O/cfg5_bench.log 2>/dev/null
cd $R
python tools/rocprof_summary.py $O/trace_cfg5 $O/trace_cfg5_summary.md | cut -c1-220 | head -14
tail -2 $O/cfg5_bench.log
find $O -name "*.db" -size +8M -delete
