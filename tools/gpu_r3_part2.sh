#!/bin/bash
# round 3, part 2: strong-scaling slices of every configuration on one GPU (machine-written: gpurun_out/slices.jsonl)
mkdir -p gpurun_out
rm -f gpurun_out/slices.jsonl
bash tools/gpu_slices.sh c3 1000000; echo "c3 slices done"
bash tools/gpu_slices.sh c2 1000000; echo "c2 slices done"
bash tools/gpu_slices.sh c4 1000000; echo "c4 slices done"
bash tools/gpu_slices.sh c5 200000; echo "c5 slices done"
