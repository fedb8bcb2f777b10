#!/bin/bash
# tile-shape experiments for the two-level kernel (tools/e2_bench.hip; variants built by hand, see the log header)
O=gpurun_out/e2b
mkdir -p $O
for v in nodefer defer auto; do
  echo "=== $v"
  timeout 120 ./tools/e2_bench_$v ${1:-512} 20
done 2>&1 | tee $O/e2_bench_${1:-512}.log
