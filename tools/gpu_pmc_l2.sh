#!/bin/bash
# L2 (TCC) hits / misses per launch of the time-loop kernel and of the Laplacian (counters in their own passes, as gpurun requires)
O=gpurun_out/r3l2
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i "TCC_HIT\|TCC_MISS\|TCC_REQ\|TCC_EA0_RDREQ\|TCC_EA0_WRREQ\|TCP_TCC_READ" | cut -c1-160 | head -20 > $R/$O/avail.txt
for c in TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $R/$O/$c -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
done
cd $R
python tools/rocprof_pmc_summary.py $O/TCC_HIT_sum $O/TCC_MISS_sum $O/TCC_REQ_sum | grep -i "euler2\|kernel |\|---" | cut -c1-260 | tee $O/summary.md
head -12 $O/avail.txt
find $O -name "*.db" -size +8M -delete
