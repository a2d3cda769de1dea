#!/bin/bash
# kernel trace of the branch-length pass over C4's four partitions (125 k-site slice), device loop (after the queue fix);
# the last 160 dispatches one by one
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$PWD}
PLLHIP_EVAL_DEVICE_NEWTON=1 TRACE_WINDOW=-160,160 tools/gpu_trace_raw.sh blo_c4_dev1_fixed python3 $R/tools/gpu_workloads.py blo_c4_125 | head -8
cat gpurun_out/traceraw_blo_c4_dev1_fixed_window.txt
