#!/bin/bash
# round 3, call 2: clean real-library drop-in run, parity of the new fp32 tiles, A/B timing of the fp32 tiles (cfg5 path),
# Infinity-Cache direction probe
O=gpurun_out/r3b
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
if [ -d _refscratch ]; then
  export PDEHIP_REFERENCE=$R/_refscratch PDEHIP_DROPIN_REAL=1 PDEHIP_DROPIN_LOG=$R/$O/dropin_outcomes.txt
  rm -f $PDEHIP_DROPIN_LOG
  echo "== drop-in tests, real library"
  timeout 1500 python -m pytest tests/test_pypde_dropin.py tests/test_pypde_plugin.py tests/test_class_pde_fuzz.py tests/test_expression_fuzz.py \
      tests/test_reference_suite.py -q -rA --tb=short -p no:cacheprovider > $O/dropin_pytest.log 2>&1
  echo "rc=$?"; tail -1 $O/dropin_pytest.log; grep "^FAILED\|^ERROR" $O/dropin_pytest.log | head -20
  grep -c "^PASSED" $PDEHIP_DROPIN_LOG; grep "^LOADED" $PDEHIP_DROPIN_LOG | sort | uniq -c
  unset PDEHIP_DROPIN_REAL PDEHIP_REFERENCE PDEHIP_DROPIN_LOG
fi
echo "== gpu tests touched by the fp32 tiles"
timeout 1500 python -m pytest tests/test_hip_euler2.py tests/test_hip_steppers.py tests/test_hip_distributed.py tests/test_baseline_configs.py tests/test_kernel_resources.py \
    -m gpu -q --tb=short -p no:cacheprovider > $O/tiles_pytest.log 2>&1
echo "rc=$?"; tail -3 $O/tiles_pytest.log; grep "^FAILED\|^ERROR" $O/tiles_pytest.log | head
echo "== fp32 tile A/B (256^3 fp32 Cahn-Hilliard: Euler step, RKF45 attempt)"
for t in "4,2,4,2" "4,2,4,1" "2,4,2,4" "2,2,2,2" "2,4,4,1" "2,2,2,4"; do
  echo "-- PDEHIP_F32_TILE=$t"; PDEHIP_F32_TILE=$t timeout 300 python tools/time_ch.py 256 100 float32 2>&1 | grep "CH n="
done | tee $O/f32_tiles_ch.log
echo "== fp32 diffusion two-step sweep"
for t in "4,2" "2,4" "2,2"; do
  for n in 256 512; do echo "-- PDEHIP_F32_TILE=$t"; PDEHIP_F32_TILE=$t timeout 300 python tools/time_euler2.py $n 200 float32 2>&1 | grep "n="; done
done | tee $O/f32_tiles_diffusion.log
echo "== Infinity Cache direction probe"
timeout 300 tools/microbench5 | tee $O/microbench5.log
