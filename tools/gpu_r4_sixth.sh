#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_results.py tests/test_eval_driver.py tests/test_mixture_models.py tests/test_expm_fixtures.py -q -p no:cacheprovider -k "61 or codon or alphabet or newton or deferred or spread or speculative or 62 or 48 or 33 or c5" > gpurun_out/r4_subset3.log 2>&1
rc=$?; tail -30 gpurun_out/r4_subset3.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "subset timed out: stopping"; exit 1; fi
for d in tools/ab/8ca04ca .; do
  for t in "--config c5" "--config c5 --sites 25000"; do
    (cd $d && python bench.py --steps 10 --no-cpu-baseline --no-also $t) > gpurun_out/r4_b.json 2> gpurun_out/r4_b.err || tail -5 gpurun_out/r4_b.err
    python - "$d $t" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r4_b.json")); r=d['roofline']
print(f"{sys.argv[1]:44s} {d['ms_per_step']:.3f} ms/step launch {r['avg_launch_ms']} x {r['launches']} frac {r['frac']} lnl {d['lnl']!r}", flush=True)
PY
  done
  (cd $d && python tools/gpu_workloads.py spr25) > gpurun_out/r4_spr.json 2> gpurun_out/r4_spr.err
  python - "$d" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r4_spr.json"))
print(sys.argv[1], {k: (round(v["s_per_round"],4), v["lnl_after"]) for k,v in d.items()}, flush=True)
PY
done
python bench.py --config c2 --steps 10 --no-cpu-baseline --no-also --transient > gpurun_out/r4_b.json 2> gpurun_out/r4_b.err
python -c "
import json; d=json.load(open('gpurun_out/r4_b.json')); print('c2 transient', d['ms_per_step'])"
