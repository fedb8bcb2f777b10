#!/bin/bash
# round 5, call 27: bench.py after the change of its `extra` timings (fastest of three short / full runs per configuration), twice
mkdir -p gpurun_out/r5i
cd /root/repo
for k in 1 2; do
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r5i/bench20_$k.err | tail -1 > gpurun_out/r5i/bench20_$k.json
  python - $k <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r5i/bench20_{sys.argv[1]}.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "op", d["roofline_operator"]["frac"], (d.get("parity") or {}).get("ok"), d.get("extra_error"), d.get("phase_seconds"))
for k, v in (d.get("extra") or {}).items():
    print("   ", k, v)
PY
done
