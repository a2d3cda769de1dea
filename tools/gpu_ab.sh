#!/bin/bash
# A/B of bench.py under environment settings inside one box: gpu_ab.sh "<bench args>" "ENV=.." "ENV=.." ...
ARGS=$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
for V in "$@"; do
  env $V python bench.py $ARGS --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - "$V" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab.json"))
print(f"{sys.argv[1]:32s} {d['value']:.4g}  {d['ms_per_step']:.3f} ms/step  frac {d['roofline']['frac']}")
PY
done; done
