#!/bin/bash
# C3 (default bench line, no CPU leg) across library revisions on ONE box: tools/ab/<sha>/ holds bench.py +
# pll-modules_amd/libpll_hip.so of that revision (built from a git worktree; boxes differ by 2-6 %).
# usage: tools/gpu_r4_ab.sh <sha> <sha> ... HEAD
mkdir -p gpurun_out
for rep in 1 2; do
for s in "$@"; do
  if [ $s = HEAD ]; then d=.; else d=tools/ab/$s; fi
  extra=""; grep -q -- "--no-also" $d/bench.py && extra="--no-also"
  out=$PWD/gpurun_out/ab_${s}_$rep
  (cd $d && python bench.py --config c3 --steps 10 --no-cpu-baseline $extra > $out.json 2> $out.err) || { echo "$s failed"; tail -3 $out.err; continue; }
  python - $s $out.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))
r=d['roofline']
print(f"{sys.argv[1]:10s} {d['ms_per_step']:.3f} ms/step  launch {r['avg_launch_ms']:.4f} ms x {r['launches']}  frac {r['frac']}", flush=True)
PY
done; done
