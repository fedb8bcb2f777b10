#!/bin/bash
# round 3, call 4: whole GPU suite, Laplacian with streaming stores, bench line
O=gpurun_out/r3d
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --maxfail=15 -rs > $O/gpu_pytest.log 2>&1
echo "rc=$?"; tail -3 $O/gpu_pytest.log; grep "^FAILED\|^ERROR" $O/gpu_pytest.log | head -20
grep "cfg5 long\|cfg5:\|cfg3 10" $O/gpu_pytest.log
echo "== Laplacian / Euler timings"
for n in 512 256; do timeout 120 python tools/time_lap.py $n 2>&1 | tail -1; done
PDEHIP_NO_NT=1 timeout 120 python tools/time_lap.py 512 2>&1 | tail -1 | sed 's/^/NO_NT /'
timeout 120 python tools/time_lap.py 512 float32 2>&1 | tail -1
timeout 200 python tools/time_ops.py 2>&1 | tail -30 > $O/time_ops.log; tail -12 $O/time_ops.log
echo "== bench"
timeout 600 python bench.py 2> $O/bench.err | tail -1 > $O/bench_n1.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3d/bench_n1.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline_operator"]["frac"], d.get("parity"), d.get("extra"), d.get("extra_error"))
PY
