#!/bin/bash
# layout experiments on physically contiguous allocations (deterministic placement): row / plane pitch paddings
for pads in "0 0" "2 0" "14 0" "62 0" "0 512" "0 2048" "0 8192" "0 16384" "0 32768" "0 65536" "0 131072" "30 4096"; do
  set -- $pads
  echo "== pad row $1 plane $2"
  timeout 100 ./tools/e2_bench_nb 512 20 1 only $1 $2 2>&1 | grep "plane pitch\|2 cells x 4\|plain" | head -3
done
