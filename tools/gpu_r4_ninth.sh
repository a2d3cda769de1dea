#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_eval_driver.py tests/test_mixture_models.py tests/test_site_repeats.py tests/test_transient.py tests/test_partition_batch.py -q -x -p no:cacheprovider > gpurun_out/r4_suite_hybrid.log 2>&1
rc=$?; echo "== plain (hybrid) rc $rc"; tail -4 gpurun_out/r4_suite_hybrid.log
if [ $rc -ne 0 ]; then exit 1; fi
env PLLHIP_TRANSIENT=1 PLLHIP_FORCED_CHILD=1 timeout -k 10 300 python -m pytest tests/test_transient.py tests/test_gpu_parity.py -q -p no:cacheprovider > gpurun_out/r4_suite_transient2.log 2>&1
echo "== forced transient (two files) rc $?"; tail -3 gpurun_out/r4_suite_transient2.log
for d in tools/ab/5a24b21 .; do
  for t in "--config c3" "--config c3 --sites 125000" "--config c4"; do
    (cd $d && python bench.py --steps 10 --no-cpu-baseline --no-also --pmc off $t) > gpurun_out/r4_b.json 2> gpurun_out/r4_b.err || tail -5 gpurun_out/r4_b.err
    python - "$d $t" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r4_b.json")); r=d['roofline']
print(f"{sys.argv[1]:44s} {d['ms_per_step']:.3f} ms/step launch {r['avg_launch_ms']} x {r['launches']} frac {r['frac']} lnl {d['lnl']!r}", flush=True)
PY
  done
  (cd $d && python tools/gpu_workloads.py alphabets alphabets32) > gpurun_out/r4_wl_$(basename $d).json 2> gpurun_out/r4_wl.err || tail -3 gpurun_out/r4_wl.err
  python - "$d" gpurun_out/r4_wl_$(basename $d).json <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))
print(sys.argv[1], {k.replace("ALPHABET_","").replace("states_","s/"): round(v["ms_per_traversal"],3) for k,v in d.items()}, flush=True)
PY
done
# 61 states on the 4x4x4 instruction without cherry folding (the three-table kernel does not spill)
for d in . tools/ab/s61x; do
  for env in "PLLHIP_S61_CHERRIES=1" "PLLHIP_S61_CHERRIES=0"; do
    (cd $d && env $env python bench.py --steps 10 --no-cpu-baseline --no-also --pmc off --config c5) > gpurun_out/r4_b.json 2> gpurun_out/r4_b.err || tail -5 gpurun_out/r4_b.err
    python - "$d $env" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r4_b.json")); r=d['roofline']
print(f"{sys.argv[1]:44s} {d['ms_per_step']:.3f} ms/step launch {r['avg_launch_ms']} x {r['launches']} frac {r['frac']}", flush=True)
PY
  done
done
