#!/bin/bash
# round 3, call 24: the tall tile (8 rows, one wave per SIMD, four plane buffers) inside the library: parity + bench A/B (PDEHIP_EULER2=8)
O=gpurun_out/r3p
mkdir -p $O
export TMPDIR=/tmp
PDEHIP_EULER2=8 timeout 900 python -m pytest tests/test_hip_euler2.py tests/test_hip_properties.py tests/test_baseline_configs.py -m gpu -q --tb=short -p no:cacheprovider --maxfail=10 -k "not fp32_tile" > $O/pytest_tall.log 2>&1
echo "rc=$?"; grep "passed\|failed" $O/pytest_tall.log | tail -1; grep "^FAILED" $O/pytest_tall.log | head
for r in 1 2 3; do
  for v in default tall; do
    if [ $v = default ]; then unset PDEHIP_EULER2; else export PDEHIP_EULER2=8; fi
    echo "-- $v"
    EXTRA=--no-extra; [ $r = 1 ] && EXTRA=
    timeout 300 python bench.py --no-cpu-baseline $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('parity'))"
    timeout 200 python tools/time_sizes.py 512x512x512 256x256x256 1024x512x512 2>&1 | grep "float64"
  done
done | tee $O/ab_tall.log
unset PDEHIP_EULER2
