#!/bin/bash
# round 4, call 1: whole GPU suite (adaptive Euler in C, stage kind 4, bc program cache, constants), bench line with repetitions
O=gpurun_out/r4a
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu (new files first)"
timeout 900 python -m pytest tests/test_hip_adaptive_euler.py tests/test_hip_frows.py -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_new.log 2>&1
echo "rc=$?"; tail -3 $O/gpu_new.log; grep "^FAILED\|^ERROR" $O/gpu_new.log | head -20
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_hip_adaptive_euler.py --deselect tests/test_hip_frows.py > $O/gpu_pytest.log 2>&1
echo "rc=$?"; tail -3 $O/gpu_pytest.log; grep "^FAILED\|^ERROR" $O/gpu_pytest.log | head -20
echo "== bench (driver arguments)"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench20.err | tail -1 > $O/bench20.json
timeout 600 python bench.py 2> $O/bench.err | tail -1 > $O/bench_n1.json
python - <<'PY'
import json
for f in ("bench20", "bench_n1"):
    d = json.load(open(f"gpurun_out/r4a/{f}.json"))
    print(f, {k: d[k] for k in ("value", "value_best", "ms_per_step")}, d["repeats"]["samples"], "frac", d["roofline"]["frac"], d["roofline"]["frac_best"],
          "op", d["roofline_operator"]["frac"], d["roofline_operator"]["frac_best"], d.get("parity"), d.get("extra_error"))
    print(json.dumps(d.get("extra"), indent=0))
PY
echo "== forced-exchange slab line (RCCL to self)"
timeout 600 python bench.py --force-distributed --steps 20 --warmup 5 --size 512 2> $O/bench_self.err | tail -1 > $O/bench_self.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4a/bench_self.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["repeats"]["samples"], d["parity"], d["per_rank"], d["roofline"]["frac"])
PY
