#!/bin/bash
# round 4, third GPU call: MFMA 4x4x4 micro-test, the whole suite (plain + the two forced modes, all failures listed), C3 A/B
mkdir -p gpurun_out
tools/micro/mfma_4x4x4 2>&1 | tee gpurun_out/r4_mfma_4x4x4.txt
timeout -k 10 600 python -m pytest tests -q -m gpu -p no:cacheprovider --ignore=tests/test_00_forced_modes.py > gpurun_out/r4_suite_plain.log 2>&1
rc=$?; tail -12 gpurun_out/r4_suite_plain.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "suite timed out: stopping"; exit 1; fi
for mode in "PLLHIP_SITE_REPEATS=2" "PLLHIP_TRANSIENT=1"; do
  env $mode PLLHIP_FORCED_CHILD=1 timeout -k 10 600 python -m pytest tests -q -m gpu -p no:cacheprovider --ignore=tests/test_00_forced_modes.py > gpurun_out/r4_suite_$mode.log 2>&1
  rc=$?; echo "== $mode rc $rc"; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r4_suite_$mode.log | tail -25
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "suite timed out: stopping"; exit 1; fi
done
tools/gpu_r4_ab.sh 6173174 HEAD 2>&1 | tee gpurun_out/r4_ab3.log
python bench.py --config c3 --steps 10 --no-cpu-baseline --no-also --transient > gpurun_out/r4_c3_transient.json 2> gpurun_out/r4_c3_transient.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_c3_transient.json")); r=d['roofline']
print("transient", f"{d['ms_per_step']:.3f} ms/step launch {r['avg_launch_ms']} x {r['launches']} frac_min {r['frac_minimum']}")
PY
