#!/bin/bash
mkdir -p gpurun_out
python tools/gpu_newton_multi_probe.py 2>&1 | tee gpurun_out/r4_newton_probe.txt
timeout -k 10 500 python -m pytest tests/test_gpu_results.py tests/test_transient.py tests/test_eval_driver.py -q -p no:cacheprovider -k "newton or deferred or spread or transient or stored or discarded or driver" > gpurun_out/r4_subset2.log 2>&1
rc=$?; tail -30 gpurun_out/r4_subset2.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "subset timed out: stopping"; exit 1; fi
for d in tools/ab/8ca04ca .; do
  for t in "--config c2" "--config c4"; do
    (cd $d && python bench.py --steps 10 --no-cpu-baseline --no-also $t) > gpurun_out/r4_b.json 2> gpurun_out/r4_b.err || tail -5 gpurun_out/r4_b.err
    python - "$d $t" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r4_b.json")); r=d['roofline']
print(f"{sys.argv[1]:28s} {d['ms_per_step']:.3f} ms/step launch {r['avg_launch_ms']} x {r['launches']} frac {r['frac']} frac_min {r['frac_minimum']}")
PY
  done
  (cd $d && python tools/gpu_workloads.py alphabets) > gpurun_out/r4_alpha.json 2> gpurun_out/r4_alpha.err
  python - "$d" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r4_alpha.json"))
print(sys.argv[1], {k: round(v["ms_per_traversal"],3) for k,v in d.items()})
PY
done
python bench.py --config c2 --steps 10 --no-cpu-baseline --no-also --transient > gpurun_out/r4_b.json 2> gpurun_out/r4_b.err
python -c "
import json; d=json.load(open('gpurun_out/r4_b.json')); print('c2 transient', d['ms_per_step'])"
