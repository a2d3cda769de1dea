#!/bin/bash
# the whole -m gpu suite on the head of the round, then the evaluate-only C3 traversal with and without the
# rate-ahead operand requests (tools/ab/pf) on the same box
mkdir -p gpurun_out
bash tools/gpu_r4_suite.sh > gpurun_out/r4_tenth_suite.txt 2>&1; tail -30 gpurun_out/r4_tenth_suite.txt
grep -q "^rc 0" gpurun_out/r4_tenth_suite.txt || echo "SUITE NOT GREEN"
for rep in 1 2; do for s in HEAD pf; do
  if [ $s = HEAD ]; then d=.; else d=tools/ab/$s; fi
  for mode in "" "--transient"; do
    out=$PWD/gpurun_out/ab10_${s}_${rep}${mode}
    (cd $d && timeout -k 10 200 python bench.py --config c3 --steps 10 --no-cpu-baseline --no-also --pmc off $mode > $out.json 2> $out.err) || { echo "$s failed"; tail -3 $out.err; exit 1; }
    python - "$s$mode" $out.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))
r=d['roofline']
print(f"{sys.argv[1]:18s} {d['ms_per_step']:.3f} ms/step  launch {r['avg_launch_ms']:.4f} ms x {r['launches']}  frac {r['frac']}", flush=True)
PY
  done
done; done
