#!/bin/bash
# 64 DNA partitions of 10 k sites under one tree: the dispatches of the last evaluation
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$PWD}
TRACE_WINDOW=-150,150 tools/gpu_trace_raw.sh manypart python3 $R/tools/gpu_many_partitions.py 4 64 10000 | head -16 | cut -c1-200
awk '{print $1, $2, $3, $4, $5, $7, $8}' gpurun_out/traceraw_manypart_window.txt | cut -c1-110
