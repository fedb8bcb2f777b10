#!/bin/bash
# round 4, call 5: complex fields on the GPU + the whole suite again (kernels back to the two-launch slab loop) + bench
O=gpurun_out/r4e
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_complex.py tests/test_hip_adaptive_euler.py -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_new.log 2>&1
echo "rc=$?"; tail -3 $O/gpu_new.log; grep "^FAILED\|^ERROR" $O/gpu_new.log | head -20
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_hip_adaptive_euler.py --deselect tests/test_hip_complex.py > $O/gpu_pytest.log 2>&1
echo "rc=$?"; tail -3 $O/gpu_pytest.log; grep "^FAILED\|^ERROR" $O/gpu_pytest.log | head -20
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench20.err | tail -1 > $O/bench20.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4e/bench20.json"))
print({k: d[k] for k in ("value", "value_best", "ms_per_step")}, d["repeats"]["samples"], "frac", d["roofline"]["frac"], d["roofline"]["frac_best"],
      "op", d["roofline_operator"]["frac"], d.get("parity"), d.get("extra_error"))
PY
