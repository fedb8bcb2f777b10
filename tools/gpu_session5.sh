#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  d=$R/gpurun_out/pmc_$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -- python $R/tools/time_euler2.py 512 20 > /dev/null 2>&1 || echo "pmc $c failed"
done
cd $R
python tools/rocprof_pmc_summary.py gpurun_out/pmc_* -o gpurun_out/pmc_euler2.md | grep -v "fillBuffer\|layout_copy" | cut -c1-250
