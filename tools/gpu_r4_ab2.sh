#!/bin/bash
# second A/B: revisions + grid sizing of the round launches (PLLHIP_ROUND_WGS) on one box
mkdir -p gpurun_out
run() { # label dir env...
  label=$1; d=$2; shift 2
  out=$PWD/gpurun_out/ab2_$label
  extra=""; grep -q -- "--no-also" $d/bench.py && extra="--no-also"
  (cd $d && env "$@" python bench.py --config c3 --steps 10 --no-cpu-baseline $extra > $out.json 2> $out.err) || { echo "$label failed"; tail -3 $out.err; return; }
  python - $label $out.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); r=d['roofline']
print(f"{sys.argv[1]:16s} {d['ms_per_step']:.3f} ms/step  launch {r['avg_launch_ms']:.4f} ms x {r['launches']}  frac {r['frac']}", flush=True)
PY
}
for rep in 1 2; do
  run 6173174 tools/ab/6173174 X=1
  run r03head tools/ab/8ca04ca X=1
  run head . X=1
  run nostore tools/ab/nostore X=1
  run head_wgs0 . PLLHIP_ROUND_WGS=0
  run head_wgs16 . PLLHIP_ROUND_WGS=16
  run head_wgs64 . PLLHIP_ROUND_WGS=64
  run nostore_wgs0 tools/ab/nostore PLLHIP_ROUND_WGS=0
  run head_trav . PLLHIP_TRAVERSE=1
done 2>&1 | tee gpurun_out/r4_ab2.log
