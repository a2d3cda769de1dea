#!/bin/bash
cd "$(dirname "$0")/.."
: > gpurun_out/manypart_r3.jsonl
for spec in "20 32 10000" "20 1 320000" "20 8 40000" "4 64 10000" "4 1 640000" "4 16 60000"; do
  python tools/gpu_many_partitions.py $spec >> gpurun_out/manypart_r3.jsonl 2>>gpurun_out/wl_r3.err
done
PLLHIP_BATCH=0 PLLHIP_EVAL_DEFERRED=0 python tools/gpu_many_partitions.py 20 32 10000 | sed 's/^{/{"variant": "per-partition launches, one wait per partition (round 2 form)", /' >> gpurun_out/manypart_r3.jsonl
cut -c1-150 gpurun_out/manypart_r3.jsonl
