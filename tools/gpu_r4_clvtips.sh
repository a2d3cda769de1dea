#!/bin/bash
# partitions without PLL_ATTRIB_PATTERN_TIP (tips as vectors, the form libpll combines with site repeats): attribute off / on,
# random and simulated data, C2 and C3
mkdir -p gpurun_out
for cfg in c2 c3; do for data in random simulated; do for rep in "" "--site-repeats"; do
  out=gpurun_out/r4_clvtips_${cfg}_${data}${rep}.json
  timeout -k 10 300 python bench.py --config $cfg --clv-tips --data $data $rep --no-also --no-cpu-baseline --pmc off --steps 10 --warmup 3 > $out 2> ${out%.json}.err || { echo "$cfg $data $rep failed"; tail -3 ${out%.json}.err; exit 1; }
  python - $out "$cfg $data clv-tips $rep" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(f"{sys.argv[2]:40s} {d['ms_per_step']:.3f} ms/step lnl {d['lnl']:.6f}", (d['config'].get('site_repeats') or {}).get('class_operations_per_step'), flush=True)
PY
done; done; done
