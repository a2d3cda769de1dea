#!/bin/bash
# round 4, first GPU call: the new transient tests, the C3 A/B across revisions, C3 evaluate-only
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_transient.py -x -q > gpurun_out/r4_transient_tests.log 2>&1
rc=$?; tail -15 gpurun_out/r4_transient_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: stopping"; exit 1; fi
tools/gpu_r4_ab.sh 6173174 cc9bbde 45140ab HEAD 2>&1 | tee gpurun_out/r4_ab.log
for t in "" "--transient"; do
  python bench.py --config c3 --steps 10 --no-cpu-baseline --no-also $t > gpurun_out/r4_c3$t.json 2> gpurun_out/r4_c3$t.err || tail -5 gpurun_out/r4_c3$t.err
  python - gpurun_out/r4_c3$t.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[1], f"{d['ms_per_step']:.3f} ms/step launch {r['avg_launch_ms']} x {r['launches']} frac {r['frac']} frac_min {r['frac_minimum']} lnl {d['lnl']!r}")
PY
done
