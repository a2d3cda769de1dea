#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/batch_modes.jsonl
: > $out
run() { PMATRIX_CALLS=per-branch python tools/gpu_many_partitions.py $3 2>>gpurun_out/batch_modes.err | sed "s/^{/{\"mode\": \"$1\", \"wgs\": \"$2\", /" | cut -c1-150 >> $out; }
for spec in "20 32 10000" "4 64 10000" "20 8 40000" "4 16 60000"; do
  PLLHIP_BATCH=0 run off - "$spec"
  PLLHIP_BATCH_MODE=0 run 0 - "$spec"
  for w in 1 2 3 4 6; do PLLHIP_BATCH_MODE=1 PLLHIP_BATCH_WGS=$w run 1 $w "$spec"; done
done
cat $out
