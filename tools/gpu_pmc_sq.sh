#!/bin/bash
# where the waves of the time-loop kernel spend their cycles: parked on s_waitcnt / issue stalls / issuing (one pass per counter)
O=gpurun_out/r3sq
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $R/$O/$c -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
done
cd $R
python tools/rocprof_pmc_summary.py $O/SQ_WAVE_CYCLES $O/SQ_WAIT_ANY $O/SQ_WAIT_INST_ANY $O/SQ_ACTIVE_INST_ANY $O/SQ_ACTIVE_INST_VALU $O/SQ_ACTIVE_INST_VMEM | grep -i "euler2\|kernel |\|---" | cut -c1-260 | tee $O/summary.md
find $O -name "*.db" -size +8M -delete
