#!/bin/bash
# quick A/B sweep of engine tuning knobs: tools/gpu_sweep.sh "ENV=.. ENV2=.." ...
mkdir -p gpurun_out
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python bench.py --config ${CFG:-c3} --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline | python tools/bench_summary.py
done
