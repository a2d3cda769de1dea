#!/bin/bash
# rows of wide tips staged in LDS when they are few: site-repeats tests, the suite under the forced attribute, then C2 / C3
# without pattern tips (attribute off / on) and the default C2 / C3 lines (their kernels: the instantiations without wide tips)
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_site_repeats.py tests/test_transient.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r4_12_a.log 2>&1; rc=$?
tail -6 gpurun_out/r4_12_a.log; [ $rc = 0 ] || exit 1
PLLHIP_SITE_REPEATS=2 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_results.py tests/test_eval_driver.py tests/test_partition_batch.py -q -m gpu -p no:cacheprovider > gpurun_out/r4_12_b.log 2>&1; rc=$?
tail -6 gpurun_out/r4_12_b.log; [ $rc = 0 ] || exit 1
for cfg in c2 c3; do for rep in "--clv-tips" "--clv-tips --site-repeats" "--clv-tips --transient" "" "--site-repeats"; do
  out="gpurun_out/r4_12_${cfg}$(echo $rep | tr -d ' ').json"
  timeout -k 10 300 python bench.py --config $cfg $rep --no-also --no-cpu-baseline --pmc off --steps 10 --warmup 3 > $out 2> ${out%.json}.err || { echo "$cfg $rep failed"; tail -3 ${out%.json}.err; exit 1; }
  python - $out "$cfg $rep" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(f"{sys.argv[2]:40s} {d['ms_per_step']:.3f} ms/step lnl {d['lnl']:.6f}", (d['config'].get('site_repeats') or {}).get('class_operations_per_step'), flush=True)
PY
done; done
