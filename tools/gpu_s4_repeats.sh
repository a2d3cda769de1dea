#!/bin/bash
# site repeats at 4 states: parity, then C2 / C4 with and without the attribute
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_site_repeats.py -x -q -m gpu > gpurun_out/s4rep_tests.log 2>&1
echo "tests exit $?"; tail -5 gpurun_out/s4rep_tests.log
timeout -k 10 300 python tests/stress_gpu_vs_oracle.py 31 > gpurun_out/s4rep_stress.log 2>&1; tail -3 gpurun_out/s4rep_stress.log
for cfg in c2 c4; do
  timeout -k 10 200 python bench.py --config $cfg --no-also --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/s4rep_${cfg}_off.json 2> gpurun_out/s4rep_${cfg}_off.err
  timeout -k 10 200 python bench.py --config $cfg --no-also --no-cpu-baseline --steps 30 --warmup 5 --site-repeats > gpurun_out/s4rep_${cfg}_on.json 2> gpurun_out/s4rep_${cfg}_on.err
  python - <<PY
import json
for k in ("off","on"):
    try:
        d=json.loads(open("gpurun_out/s4rep_${cfg}_%s.json"%k).read().strip().splitlines()[-1])
        print("$cfg",k,d["ms_per_step"],d["roofline"].get("frac"),d["roofline"].get("frac_minimum"),d.get("site_repeats"))
    except Exception as ex: print("$cfg",k,"failed",ex)
PY
done
