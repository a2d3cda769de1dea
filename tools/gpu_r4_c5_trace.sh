#!/bin/bash
# the 61-state traversal at the 25 k-site slice, dispatch by dispatch (the last two steps)
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$PWD}
TRACE_WINDOW=-64,64 tools/gpu_trace_raw.sh c5_25k python3 $R/bench.py --config c5 --sites 25000 --steps 6 --warmup 2 --no-cpu-baseline --no-also --pmc off | head -12
cat gpurun_out/traceraw_c5_25k_window.txt
