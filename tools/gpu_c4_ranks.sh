#!/bin/bash
# C4 on N GPUs, rehearsed on one: every rank's share under the cost-balanced assignment (bench.py --as-rank R/N) and
# under the uniform 1/N split of every partition; the slowest rank sets the step.  Appends to gpurun_out/c4_ranks.jsonl
cd "$(dirname "$0")/.."
N=${1:-8}
out=gpurun_out/c4_ranks.jsonl
: > $out
for mode in balanced uniform; do
  for ((r = 0; r < N; ++r)); do
    if [ $mode = uniform ] && [ $r -gt 0 ]; then continue; fi      # every rank holds the same shares
    PLLHIP_BENCH_BALANCE=$([ $mode = balanced ] && echo 1 || echo 0) python bench.py --config c4 --as-rank $r/$N --steps 20 --warmup 3 --no-cpu-baseline 2>>gpurun_out/c4_ranks.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'assignment': '$mode', 'rank': $r, 'of': $N, 'ms_per_step': d['ms_per_step'], 'sites_on_rank': d['config']['sites_per_gpu'],
                  'partitions': [p for p in d['config']['partitions'] if not p.get('remote')], 'launches_per_step': d['config']['partial_launches_per_step']}))" >> $out
  done
done
python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline 2>>gpurun_out/c4_ranks.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'assignment': 'one GPU, everything', 'ms_per_step': d['ms_per_step']}))" >> $out
cat $out
