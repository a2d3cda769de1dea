#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/round_wgs2.jsonl
: > $out
for k in 0 2 4 6 8 12 16 32; do
  export PLLHIP_ROUND_WGS=$k
  for spec in "20 32 10000" "4 64 10000" "20 8 40000"; do
    PMATRIX_CALLS=per-branch python tools/gpu_many_partitions.py $spec 2>>gpurun_out/round_wgs.err | sed "s/^{/{\"round_wgs\": $k, /" | cut -c1-140 >> $out || exit 1
  done
done
cat $out
