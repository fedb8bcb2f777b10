#!/bin/bash
# round 4, final build (open rows of up to eight columns): the whole GPU suite, smoke(), the bench line with driver arguments
O=gpurun_out/r4final4
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_all.log 2>&1
echo "rc=$?"; grep "passed\|failed" $O/gpu_all.log | tail -2; grep "^FAILED\|^ERROR" $O/gpu_all.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench20.err | tail -1 > $O/bench20.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4final4/bench20.json"))
print({k: d[k] for k in ("value", "value_best", "ms_per_step")}, "frac", d["roofline"]["frac"], "op", d["roofline_operator"]["frac"], d["roofline"]["copy_ceiling"], d["parity"]["ok"])
for k, v in d["extra"].items(): print("   ", k, v.get("us_per_step"), v.get("us_per_attempt"))
PY
