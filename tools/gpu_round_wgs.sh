#!/bin/bash
# sweep PLLHIP_ROUND_WGS (workgroups per CU a round launch aims for) over the workloads it matters for
cd "$(dirname "$0")/.."
out=gpurun_out/round_wgs.jsonl
: > $out
for k in 0 2 4 8 16; do
  export PLLHIP_ROUND_WGS=$k
  for spec in "20 32 10000" "4 64 10000"; do
    PMATRIX_CALLS=per-branch python tools/gpu_many_partitions.py $spec 2>>gpurun_out/round_wgs.err | sed "s/^{/{\"round_wgs\": $k, /" >> $out || exit 1
  done
  for cfg in "c3 --sites 125000" "c3 --sites 250000" "c3" "c2 --sites 125000" "c2" "c4" "c4 --sites 125000"; do
    python bench.py --config $cfg --steps 10 --no-cpu-baseline 2>>gpurun_out/round_wgs.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'round_wgs': $k, 'cfg': '$cfg', 'ms_per_step': d['ms_per_step'], 'launches_per_step': d['config']['partial_launches_per_step']}))" >> $out || exit 1
  done
  echo "done k=$k" >&2
done
cat $out
