#!/bin/bash
# round 4, fourth GPU call: parity subset with the 4x4x4 tail + the multi-partition Newton loop, C3 A/B, benches, then the plain suite with durations
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_site_repeats.py tests/test_transient.py tests/test_eval_driver.py tests/test_gpu_results.py -q -x -p no:cacheprovider -k "20 or golden or chained or schedules or newton" > gpurun_out/r4_subset.log 2>&1
rc=$?; tail -25 gpurun_out/r4_subset.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "subset timed out: stopping"; exit 1; fi
tools/gpu_r4_ab.sh 6173174 HEAD 2>&1 | tee gpurun_out/r4_ab4.log
for t in "--transient" "--config c2" "--config c2 --transient" "--config c4"; do
  python bench.py --steps 10 --no-cpu-baseline --no-also $t > gpurun_out/r4_b.json 2> gpurun_out/r4_b.err || tail -5 gpurun_out/r4_b.err
  python - "$t" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r4_b.json")); r=d['roofline']
print(f"{sys.argv[1]:28s} {d['ms_per_step']:.3f} ms/step launch {r['avg_launch_ms']} x {r['launches']} frac {r['frac']} frac_min {r['frac_minimum']} share {r['kernel_share_of_step']} fam {r.get('families')}")
PY
done
timeout -k 10 600 python -m pytest tests -q -m gpu -p no:cacheprovider --ignore=tests/test_00_forced_modes.py --durations=40 > gpurun_out/r4_suite_plain.log 2>&1
rc=$?; tail -60 gpurun_out/r4_suite_plain.log
