#!/bin/bash
# round 3, call 3: whole GPU suite (fp32 tiles fixed, BC programs, C loops), A/B of non-temporal loads / stores, bench line
O=gpurun_out/r3c
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $O/gpu_pytest.log 2>&1
echo "rc=$?"; tail -3 $O/gpu_pytest.log; grep "^FAILED\|^ERROR" $O/gpu_pytest.log | head
echo "== A/B: non-temporal loads / stores (two interleaved rounds)"
for r in 1 2; do
  for v in default ntl ntl2 nts ntls; do
    if [ $v = default ]; then unset PDEHIP_LIB; else export PDEHIP_LIB=$R/tools/variants/libpdehip_$v.so; fi
    echo -n "$v: "; timeout 120 python tools/time_euler2.py 512 200 2>&1 | tail -1
    timeout 120 python tools/time_lap.py 512 2>&1 | tail -1
  done
done | tee $O/ab_nt.log
unset PDEHIP_LIB
for v in default ntl ntls; do
  if [ $v = default ]; then unset PDEHIP_LIB; else export PDEHIP_LIB=$R/tools/variants/libpdehip_$v.so; fi
  echo -n "$v 256^3: "; timeout 120 python tools/time_euler2.py 256 300 2>&1 | tail -1
  echo -n "$v 64x512x512: "; timeout 120 python tools/time_euler2.py 64,512,512 300 2>&1 | tail -1
  timeout 120 python tools/time_lap.py 256 2>&1 | tail -1
  echo -n "$v fp32 CH rkf45: "; timeout 120 python tools/time_ch.py 256 100 float32 2>&1 | tail -1
done | tee -a $O/ab_nt.log
unset PDEHIP_LIB
echo "== bench"
timeout 600 python bench.py 2> $O/bench.err | tail -1 > $O/bench_n1.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3c/bench_n1.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d.get("roofline_operator"), d.get("parity"), d.get("extra"), d.get("extra_error"))
PY
