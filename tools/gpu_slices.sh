#!/bin/bash
# strong-scaling proxy on one GPU: the per-GPU slice of an N-way split of a configuration
CFG=${1:-c3}
mkdir -p gpurun_out
for N in 1000000 500000 250000 125000; do
  python bench.py --config $CFG --sites $N --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/slice.json 2> gpurun_out/slice.err
  python - $N <<'PY'
import json,sys
d=json.load(open("gpurun_out/slice.json"))
print(f"sites {sys.argv[1]:>8s}  {d['value']:.4g} site-updates/s  {d['ms_per_step']:.3f} ms/step  frac {d['roofline']['frac']}  kernel share {d['roofline']['kernel_share_of_step']}  launches {d['roofline']['launches']//d['steps']}")
PY
done
