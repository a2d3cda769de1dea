#!/bin/bash
mkdir -p gpurun_out
env PLLHIP_TRANSIENT=1 PLLHIP_FORCED_CHILD=1 timeout -k 10 500 python -m pytest tests -q -m gpu -p no:cacheprovider --ignore=tests/test_00_forced_modes.py > gpurun_out/r4_suite_transient.log 2>&1
rc=$?; echo "== forced transient rc $rc"; grep -E "^FAILED|^ERROR| passed| failed" gpurun_out/r4_suite_transient.log | tail -25
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "suite timed out: stopping"; exit 1; fi
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_eval_driver.py tests/test_gpu_results.py tests/test_mixture_models.py tests/test_site_repeats.py tests/test_transient.py tests/test_partition_batch.py -q -x -p no:cacheprovider > gpurun_out/r4_suite_4x4.log 2>&1
rc=$?; echo "== plain (4x4x4 everywhere) rc $rc"; tail -8 gpurun_out/r4_suite_4x4.log
if [ $rc -ne 0 ]; then exit 1; fi
for d in tools/ab/5a24b21 .; do
  for t in "--config c3" "--config c3 --transient" "--config c3 --sites 125000" "--config c4"; do
    (cd $d && python bench.py --steps 10 --no-cpu-baseline --no-also --pmc off $t) > gpurun_out/r4_b.json 2> gpurun_out/r4_b.err || tail -5 gpurun_out/r4_b.err
    python - "$d $t" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r4_b.json")); r=d['roofline']
print(f"{sys.argv[1]:44s} {d['ms_per_step']:.3f} ms/step launch {r['avg_launch_ms']} x {r['launches']} frac {r['frac']} lnl {d['lnl']!r}", flush=True)
PY
  done
  (cd $d && python tools/gpu_workloads.py alphabets alphabets32 w2 blo125 blo_c4) > gpurun_out/r4_wl_$(basename $d).json 2> gpurun_out/r4_wl.err || tail -3 gpurun_out/r4_wl.err
  python - "$d" gpurun_out/r4_wl_$(basename $d).json <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))
for k,v in d.items():
    print(sys.argv[1], k, {kk: (round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("ms_per_traversal","us_per_derivative_call","us_per_derivative_call_incl_everything","s_per_smoothing_pass","newton_iterations","lnl_after","lnl","us_per_call")}, flush=True)
PY
done
