#!/bin/bash
# SQ_INSTS_VALU / SQ_WAVES per launch of the time-loop kernel with and without the unit-spacing instance (VERDICT r1 #5 asks for
# the VALU count before / after a change to euler2_kernel); counters in their own passes, as gpurun requires
O=gpurun_out/r2valu
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
for v in unit nounit; do
  if [ $v = nounit ]; then export PDEHIP_NO_UNIT=1; else unset PDEHIP_NO_UNIT; fi
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace -d $R/$O/valu_$v -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES --kernel-trace -d $R/$O/busy_$v -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
for v in unit nounit; do echo "## $v"; python tools/rocprof_pmc_summary.py $O/valu_$v $O/busy_$v | grep -i "euler2\|kernel |\|---" | cut -c1-260; done | tee $O/summary.md
find $O -name "*.db" -size +8M -delete
