#!/bin/bash
# many-partition probe: per-branch vs batched P-matrix calls, and the single-partition rate to compare with
cd "$(dirname "$0")/.."
out=gpurun_out/manypart_probe.jsonl
: > $out
for spec in "20 32 10000" "20 1 320000" "4 64 10000" "4 1 640000"; do
  for mode in per-branch batched; do
    PMATRIX_CALLS=$mode python tools/gpu_many_partitions.py $spec >> $out 2>gpurun_out/manypart_probe.err || exit 1
  done
done
cat $out
