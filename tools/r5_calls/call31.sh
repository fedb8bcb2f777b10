#!/bin/bash
# round 5, call 31: the whole GPU suite + smoke() + the bench line on the library with the block-loop schedule change (next interior sweep released behind the pack launch)
mkdir -p gpurun_out/r5l
cd /root/repo
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r5l/gpu_all.log 2>&1
echo "rc=$?"; tail -3 gpurun_out/r5l/gpu_all.log; grep "^FAILED\|^ERROR" gpurun_out/r5l/gpu_all.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r5l/bench20.err | tail -1 > gpurun_out/r5l/bench20.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5l/bench20.json"))
print({k: d[k] for k in ("value", "value_best", "ms_per_step")}, "frac", d["roofline"]["frac"], "op", d["roofline_operator"]["frac"], (d.get("parity") or {}).get("ok"), d.get("phase_seconds"))
for k, v in (d.get("extra") or {}).items():
    print("   ", k, v)
PY
