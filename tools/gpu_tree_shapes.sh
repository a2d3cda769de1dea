#!/bin/bash
# the benchmark configurations on other tree shapes (the schedule forms were tuned on random trees): gpurun_out/tree_shapes.jsonl
cd "$(dirname "$0")/.."
out=gpurun_out/tree_shapes.jsonl
: > $out
for cfg in c3 c2 "c3 --sites 125000" "c3 --site-repeats"; do
  for shape in random ladder balanced; do
    python bench.py --config $cfg --tree $shape --steps 10 --no-cpu-baseline 2>>gpurun_out/tree_shapes.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'config': '$cfg', 'tree': '$shape', 'ms_per_step': d['ms_per_step'], 'site_updates_per_s': d['value'], 'frac': r['frac'], 'frac_minimum': r['frac_minimum'],
                  'launches_per_step': d['config']['partial_launches_per_step'], 'lnl': d['lnl'], 'site_repeats': d['config']['site_repeats']}))" >> $out
  done
done
cat $out
