#!/bin/bash
# C5 step time under the launch-shape switches of the 61-state family
mkdir -p gpurun_out
for V in "" "PLLHIP_S61_GXMUL=2" "PLLHIP_S61_GXMUL=4" "PLLHIP_S61_RATEPAR=0" "PLLHIP_S61_RATEPAR=0 PLLHIP_S61_GXMUL=2" "PLLHIP_S61_ALAP=0"; do
  env $V python bench.py --config c5 --no-cpu-baseline --steps 10 > gpurun_out/c5v.json 2> gpurun_out/c5v.err
  python - "$V" <<'PY'
import json,sys
d=json.load(open("gpurun_out/c5v.json"))
print(f"{sys.argv[1]:45s} ms/step {d['ms_per_step']:.3f}  TFLOP/s {d['roofline']['achieved']}  frac {d['roofline']['frac']}")
PY
done
