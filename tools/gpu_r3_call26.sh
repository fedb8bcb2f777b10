#!/bin/bash
# round 3, call 26: library with ABI 4 (expression conditions on slabs / blocks): full GPU suite, smoke(), the real py-pde against it
# (reference shipped as scratch), the slab worker with its differential fuzz, cost of refreshed conditions at 512^3
O=gpurun_out/r3i
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --maxfail=40 > $O/pytest_gpu_final.log 2>&1
echo "rc=$?"; grep "passed\|failed" $O/pytest_gpu_final.log | tail -1; grep "^FAILED" $O/pytest_gpu_final.log | head -40
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python tools/time_bc_program.py 512 100 > $O/time_bc_program.log 2>&1; grep BCPROG $O/time_bc_program.log || tail -5 $O/time_bc_program.log
export PDEHIP_REFERENCE=$R/_refscratch PDEHIP_DROPIN_REAL=1 PDEHIP_DROPIN_LOG=$R/$O/dropin_outcomes.txt
rm -f $PDEHIP_DROPIN_LOG
timeout 1800 python -m pytest tests/test_pypde_dropin.py tests/test_pypde_plugin.py tests/test_class_pde_fuzz.py tests/test_expression_fuzz.py \
    tests/test_reference_suite.py -q -rA --tb=short -p no:cacheprovider > $O/dropin_pytest.log 2>&1
echo "rc=$?"; tail -1 $O/dropin_pytest.log; grep "^FAILED\|^ERROR" $O/dropin_pytest.log | head -20
grep -c "^PASSED" $PDEHIP_DROPIN_LOG; grep "^LOADED" $PDEHIP_DROPIN_LOG | sort | uniq -c
for d in slab auto; do
  PDEHIP_WORKER_FUZZ=18 PDEHIP_WORKER_DECOMPOSITION=$d timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29611 tests/pypde_slab_worker.py > $O/slab_worker_$d.log 2>&1
  echo "worker $d rc=$?"; grep PYPDESLAB $O/slab_worker_$d.log | tail -1 | cut -c1-200; grep -o '"failures": \[[^]]*\]' $O/slab_worker_$d.log | tail -1
done
