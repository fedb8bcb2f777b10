#!/bin/bash
# round 3, call 21: tensor-field states on the real library (mirror + oracle), and through the real py-pde
O=gpurun_out/r3o
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_expressions.py tests/test_expression_fuzz_gpu.py tests/test_hip_frows.py -m gpu -q --tb=short -p no:cacheprovider --maxfail=20 > $O/pytest.log 2>&1
echo "rc=$?"; grep "passed\|failed" $O/pytest.log | tail -1; grep "^FAILED" $O/pytest.log | head -20
if [ -d _refscratch ]; then
  PDEHIP_REFERENCE=$R/_refscratch PDEHIP_DROPIN_REAL=1 timeout 900 python -m pytest tests/test_pypde_dropin.py -q --tb=short -p no:cacheprovider -k "tensor_fields or nonlinearly or user_funcs or vector" 2>&1 | tail -4
fi
