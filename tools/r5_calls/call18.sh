#!/bin/bash
# round 5, call 18: tall tile by default for large fields (parity at 512^3 + bench twice: is the slow cfg2 of a first process a per-box one-off?),
# then the real py-pde against the real library (tools/gpu_r5_dropin.sh)
mkdir -p gpurun_out/r5d
cd /root/repo
timeout 900 python -m pytest tests/test_baseline_configs.py tests/test_hip_euler2.py tests/test_hip_properties.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r5d/bench20_$i.json
python - <<PY
import json
d=json.load(open("gpurun_out/r5d/bench20_$i.json"))
print("bench20 run $i", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "op", d["roofline_operator"]["frac"], d["roofline"]["kernel"][:40], (d.get("parity") or {}).get("ok"), d.get("extra_error"))
print("   vector_laplace", d["roofline_operators"]["vector_laplace"]["frac"], "cfg2", d["extra"]["cfg2_diffusion_1024sq_f64_euler"], d["phase_seconds"])
PY
done
bash tools/gpu_r5_dropin.sh
