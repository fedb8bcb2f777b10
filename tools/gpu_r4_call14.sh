#!/bin/bash
# round 4, call 14: open rows of up to four columns, complex axis derivatives / gradient_squared, bench line with the new extra entry
O=gpurun_out/r4n
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_euler2.py tests/test_hip_complex.py tests/test_hip_frows.py tests/test_expression_fuzz_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/gpu_tests.log 2>&1
echo "rc=$?"; tail -6 $O/gpu_tests.log
timeout 300 python tools/time_sizes.py 515x515x515 512x512x516 513x513x513 2>/dev/null | tee $O/sizes_open4.log | grep "^| 5" | cut -c1-110
PDEHIP_OPEN_ROWS=0 timeout 300 python tools/time_sizes.py 515x515x515 512x512x516 2>/dev/null | tee $O/sizes_closed4.log | grep "^| 5" | cut -c1-110
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4n/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline_operator"]["frac"], d["parity"])
print(json.dumps(d["extra"], indent=0)[:1500])
PY
