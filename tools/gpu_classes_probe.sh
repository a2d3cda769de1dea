#!/bin/bash
# site repeats, second step: parity, stress sweep, then C2 / C3 / C4 with the attribute on random and simulated data
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_site_repeats.py -x -q -m gpu > gpurun_out/cls_tests.log 2>&1
echo "tests exit $?"; tail -15 gpurun_out/cls_tests.log
timeout -k 10 400 python tests/stress_gpu_vs_oracle.py 91 > gpurun_out/cls_stress.log 2>&1; tail -3 gpurun_out/cls_stress.log
for cfg in c2 c3 c4; do
  for data in random simulated; do
    timeout -k 10 300 python bench.py --config $cfg --no-also --no-cpu-baseline --steps 20 --warmup 4 --site-repeats --data $data > gpurun_out/cls_${cfg}_${data}.json 2> gpurun_out/cls_${cfg}_${data}.err
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/cls_${cfg}_${data}.json").read().strip().splitlines()[-1])
    print("$cfg $data", round(d["ms_per_step"],3), d.get("site_repeats"))
except Exception as ex: print("$cfg $data failed", ex)
PY
  done
done
