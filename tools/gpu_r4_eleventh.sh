#!/bin/bash
# tips kept per class of sites whether or not the partition asks for site repeats: the whole suite, then C2 / C3 / C4 without
# pattern tips (attribute off / on)
mkdir -p gpurun_out
bash tools/gpu_r4_suite.sh > gpurun_out/r4_eleventh_suite.txt 2>&1; tail -12 gpurun_out/r4_eleventh_suite.txt
grep -q "^rc 0" gpurun_out/r4_eleventh_suite.txt || { echo "SUITE NOT GREEN"; exit 1; }
for cfg in c2 c3 c4; do for rep in "" "--site-repeats" "--transient"; do
  out=gpurun_out/r4_clvtips2_${cfg}${rep}.json
  timeout -k 10 300 python bench.py --config $cfg --clv-tips $rep --no-also --no-cpu-baseline --pmc off --steps 10 --warmup 3 > $out 2> ${out%.json}.err || { echo "$cfg $rep failed"; tail -3 ${out%.json}.err; exit 1; }
  python - $out "$cfg clv-tips $rep" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(f"{sys.argv[2]:40s} {d['ms_per_step']:.3f} ms/step lnl {d['lnl']:.6f}", (d['config'].get('site_repeats') or {}).get('class_operations_per_step'), flush=True)
PY
done; done
