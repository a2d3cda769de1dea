#!/bin/bash
# strong-scaling proxy on one GPU: the per-GPU slice of a 1 / 2 / 4 / 8-way split of a configuration
# (tools/gpu_slices.sh <cfg> [full site count]); appends one JSON object per slice to gpurun_out/slices.jsonl
CFG=${1:-c3}
FULL=${2:-1000000}
mkdir -p gpurun_out
for DIV in 1 2 4 8; do
  N=$((FULL / DIV))
  python bench.py --config $CFG --sites $N --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/slice.json 2> gpurun_out/slice.err
  python - $CFG $N $DIV <<'PY'
import json,sys
d=json.load(open("gpurun_out/slice.json"))
r=d['roofline']
print(f"{sys.argv[1]} sites {sys.argv[2]:>8s} (1/{sys.argv[3]})  {d['value']:.4g} site-updates/s  {d['ms_per_step']:.3f} ms/step  frac {r['frac']}  kernel share {r['kernel_share_of_step']}  launches {r['launches']//d['steps']}")
with open("gpurun_out/slices.jsonl","a") as f:
    f.write(json.dumps({"config":sys.argv[1],"sites":int(sys.argv[2]),"share_of_full":f"1/{sys.argv[3]}","site_updates_per_s":d['value'],"ms_per_step":d['ms_per_step'],"roofline_frac":r['frac'],"kernel_share_of_step":r['kernel_share_of_step'],"partial_launches_per_step":r['launches']//d['steps']})+"\n")
PY
done
