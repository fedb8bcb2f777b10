#!/bin/bash
# round 5: the REAL py-pde (shipped as git-ignored scratch, removed after the call) against the real libpdehip.so on the MI355X -
# all drop-in files incl. this round's backend methods (products, expression functions, ghost-cell setters) on the 128-byte row layout
O=gpurun_out/r5dropin
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
export PDEHIP_REFERENCE=$R/_refscratch PDEHIP_DROPIN_REAL=1 PDEHIP_DROPIN_LOG=$R/$O/dropin_outcomes.txt
rm -f $PDEHIP_DROPIN_LOG
timeout 2400 python -m pytest tests/test_pypde_dropin.py tests/test_pypde_plugin.py tests/test_class_pde_fuzz.py tests/test_expression_fuzz.py \
    tests/test_reference_suite.py tests/test_complex.py tests/test_adaptive_euler.py tests/test_device_hooks.py -q -rA --tb=short -p no:cacheprovider -m "not gpu" > $O/dropin_pytest.log 2>&1
echo "rc=$?"; tail -1 $O/dropin_pytest.log; grep "^FAILED\|^ERROR" $O/dropin_pytest.log | head -30
grep -c "^PASSED" $PDEHIP_DROPIN_LOG; grep "^LOADED" $PDEHIP_DROPIN_LOG | sort | uniq -c
grep "^PASSED\|^FAILED\|^SKIPPED" $O/dropin_pytest.log | sed 's/::.*//' | sort | uniq -c
