#!/bin/bash
# round 3, call 5: block decomposition on the device (self exchange), whole suite, slab vs block probes
O=gpurun_out/r3e
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --maxfail=15 > $O/gpu_pytest.log 2>&1
echo "rc=$?"; grep "passed\|failed" $O/gpu_pytest.log | tail -2; grep "^FAILED\|^ERROR" $O/gpu_pytest.log | head -20
echo "== probes: halo to self through RCCL, one GPU"
timeout 300 python tools/probe_slab.py 64,512,512 300 2>&1 | grep "exchange=\|euler_run" | tee $O/probe_slab.log
timeout 300 python tools/probe_block.py 256,256,256 300 2>&1 | grep "exchange=" | tee $O/probe_block.log
timeout 300 python tools/probe_block.py 128,128,128 300 2>&1 | grep "exchange=" | tee -a $O/probe_block.log
