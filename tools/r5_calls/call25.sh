#!/bin/bash
# round 5, call 25: the tall tile with the ragged-row code (variant library, PDEHIP_TALL_RAGGED) - parity against the oracle on odd shapes, then
# A/B timing at sizes that are not multiples of the tile; then the real py-pde against the real library (tools/gpu_r5_dropin.sh)
mkdir -p gpurun_out/r5g
cd /root/repo
V=$PWD/tools/variants/libpdehip_tallr.so
PDEHIP_ALLOW_LIB_OVERRIDE=1 PDEHIP_LIB=$V PDEHIP_EULER2=8 timeout 600 python tools/check_tall_ragged.py 2>&1 | tail -12 | tee gpurun_out/r5g/check_tall_ragged.log
PDEHIP_EULER2=8 timeout 600 python tools/check_tall_ragged.py 2>&1 | tail -3 | sed 's/^/(in-tree build, forced 8: tall where rows fit) /'
L=gpurun_out/r5g/ab_tall_ragged.log
: > $L
for rep in 1 2; do
  echo "== in-tree build" >> $L
  timeout 300 python tools/time_sizes.py 513x513x513 511x511x511 510x510x510 512x512x520 500x500x300 600x520x520 2>/dev/null | grep "float64" >> $L
  echo "== variant (tall tile with ragged rows)" >> $L
  PDEHIP_ALLOW_LIB_OVERRIDE=1 PDEHIP_LIB=$V timeout 300 python tools/time_sizes.py 513x513x513 511x511x511 510x510x510 512x512x520 500x500x300 600x520x520 2>/dev/null | grep "float64" >> $L
done
cat $L
bash tools/gpu_r5_dropin.sh
