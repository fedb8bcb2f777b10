#!/bin/bash
# round 4, call 17: where the waves of shell_kernel spend their cycles (one counter per pass); tails with the split of <= 2 columns
O=gpurun_out/r4q
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_hip_tails.py tests/test_hip_operators.py tests/test_hip_frows.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/tests.log 2>&1
echo "tests rc=$?"; tail -2 $O/tests.log
cd /tmp
CS="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
for c in $CS; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $R/$O/$c -- python $R/tools/run_bc_program.py 512 20 > /dev/null 2>&1
done
cd $R
args=""; for c in $CS; do args="$args $O/$c"; done
python tools/rocprof_pmc_summary.py $args | grep -i "shell\|kernel |\|---" | cut -c1-300 | tee $O/shell_counters.md
find $O -name "*.db" -size +8M -delete
timeout 300 python tools/time_sizes.py 513x513x513 514x514x514 515x515x515 512x512x516 2>/dev/null | tee $O/sizes.log | grep "^| 5" | cut -c1-110
