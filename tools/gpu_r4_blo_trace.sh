#!/bin/bash
# where a branch-length pass over C4's four partitions (125 k-site slice) spends its time: device loop / host loop, kernel trace
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$PWD}
for dev in 1 0; do
  PLLHIP_EVAL_DEVICE_NEWTON=$dev python tools/gpu_workloads.py blo_c4_125 > gpurun_out/r4_blo_c4_dev$dev.json 2> gpurun_out/r4_blo.err || tail -3 gpurun_out/r4_blo.err
  python - $dev <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r4_blo_c4_dev{sys.argv[1]}.json"))["BLO_c4_125000"]
print("device_newton", sys.argv[1], {k: d[k] for k in ("s_per_smoothing_pass","newton_iterations","us_per_derivative_call_incl_everything","lnl_after")}, flush=True)
PY
done
PLLHIP_EVAL_DEVICE_NEWTON=1 tools/gpu_trace_raw.sh blo_c4_dev1 python3 $R/tools/gpu_workloads.py blo_c4_125 | head -14
